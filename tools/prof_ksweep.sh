#!/bin/bash
# dev: rocprofv3 kernel trace of SparseVI's weight optimisation at one coreset size (tools/c5_ksweep.py).
# usage: tools/prof_ksweep.sh TAG K [extra args of c5_ksweep.py]    -> gpurun_out/prof_ks_TAG_kK.txt
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; K=$2; shift 2
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/raw_ks_$K
rocprofv3 --kernel-trace --stats -d $O/raw_ks_$K -o t -- python $R/tools/c5_ksweep.py --ks $K --reps 2 --rows 200000 "$@" > $O/prof_ks_${TAG}_k$K.json 2> $O/prof_ks_${TAG}_k$K.err
python $R/tools/rocpd_summary.py $(find $O/raw_ks_$K -name "*.db" | head -1) > $O/prof_ks_${TAG}_k$K.txt
rm -rf $O/raw_ks_$K
head -25 $O/prof_ks_${TAG}_k$K.txt
