#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
tools/profile_round.sh r03 c5 c5pmc fam shards > gpurun_out/prof_r03_b.log 2>&1; tail -3 gpurun_out/prof_r03_b.log
O=gpurun_out/prof_r03
python tools/omp_hist.py --rows 1000000 --itrs 140 > $O/omp_hist_c3.txt 2>&1
python tools/omp_hist.py --rows 1000000 --itrs 140 --randn --quiet > $O/omp_hist_randn.txt 2>&1
tail -3 $O/omp_hist_randn.txt
