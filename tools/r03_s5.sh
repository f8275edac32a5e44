#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s5; mkdir -p $O
timeout 600 python tools/proj_ab.py > $O/proj_ab.txt 2>&1; cat $O/proj_ab.txt | tail -10
for th in 1024 512 256; do
  BCX_OMP_THREADS=$th timeout 600 python tools/omp_hist.py --rows 1000000 --itrs 140 > $O/omp_hist_c3_t$th.log 2>&1
  echo "== threads $th"; sed -n '/^it  60/,/^it  61/p' $O/omp_hist_c3_t$th.log | cut -c1-420; tail -6 $O/omp_hist_c3_t$th.log
done
