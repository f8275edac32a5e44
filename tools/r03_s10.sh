#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); O=$R/gpurun_out/s10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for fam in logistic poisson; do
rm -rf $O/raw
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/raw -o p -- python $R/tools/proj_shape.py --family $fam --mode colsum --dim 300 --reps 8 > $O/${fam}_pmc.out 2> $O/${fam}_pmc.err
python $R/tools/rocpd_summary.py $(find $O/raw -name "*.db" | head -1) > $O/${fam}_colsum_pmc.txt
grep "proj_kernel" $O/${fam}_colsum_pmc.txt | cut -c1-60,90-170
done
rm -rf $O/raw
