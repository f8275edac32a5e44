#!/bin/bash
# round-3 GPU session 4: OMP fine stamps; SELECT rewrite + Poisson log series; family x mode projection table
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s4; mkdir -p $O
run() { local name=$1; shift; timeout 1500 "$@" > $O/$name.log 2>&1; echo "$name rc=$? :: $(tail -n 1 $O/$name.log)" | tee -a $O/summary.txt; }
run omp_hist_c3 python tools/omp_hist.py --rows 1000000 --itrs 140
sed -n '/^it  60/,/^it  64/p' $O/omp_hist_c3.log | cut -c1-400
tail -7 $O/omp_hist_c3.log
run proj python -m pytest tests/test_gpu_projection.py -q -p no:cacheprovider
run svi_sharded python -m pytest tests/test_gpu_sharded.py -q -p no:cacheprovider -k "sparsevi"
run fullsize5 python -m pytest tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "config5"
for fam in linreg logistic poisson; do for mode in colsum select write write_raw; do
  timeout 300 python tools/proj_shape.py --family $fam --mode $mode --dim 300 --reps 20 2>/dev/null | tail -1 | tee -a $O/proj_table.txt
done; done
