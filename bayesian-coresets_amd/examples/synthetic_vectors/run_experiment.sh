#!/usr/bin/env bash
# Both sweeps of the synthetic-vector experiment (normal: N=10k, d=100, 50 log-spaced sizes up to 1000;
# axis: N=100 unit vectors, 10 sizes up to 100), five seeds, every algorithm.  Results land in results/
# (one arg-hashed CSV per run, see ../common/results.py); summarise with e.g.
#   python3 main.py plot Ms err --summarize trial --groupby Ms --plot_legend alg
set -euo pipefail
cd "$(dirname "$0")"

sweep() {   # sweep <data_type> [extra main.py arguments...]
  local kind=$1; shift
  local seed method
  for seed in $(seq 1 5); do
    for method in GIGA FW OMP US; do
      python3 main.py --data_type "$kind" --trial "$seed" --alg "$method" "$@" run
    done
  done
}

sweep normal
sweep axis --data_num 100 --coreset_size_max 100 --coreset_num_sizes 10
