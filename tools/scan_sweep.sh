#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
# dev tool: sweep scan-kernel launch parameters on the GPU box
cd $GRAFT_REPO_ROOT/bayesian-coresets_amd
for v in "8 4" "16 4" "16 8" "4 4" "32 8"; do
  set -- $v
  touch csrc/scan.hip
  make EXTRA="-DBCX_LOADS_IN_FLIGHT=$1 -DBCX_UR_MAX=$2" >/dev/null 2>&1
  for g in 1024 2048 4096; do
    echo "== loads=$1 urmax=$2 grid=$g"
    BCX_SCAN_GRID=$g python ../tools/gpu_quick.py sweep 2>&1 | grep "alg="
  done
done
touch csrc/scan.hip; make >/dev/null 2>&1
