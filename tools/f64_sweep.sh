#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
cd $GRAFT_REPO_ROOT
for g in 256 512 1024 2048; do
  echo "== grid cap $g"
  BCX_SCAN_GRID=$g python - <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "bayesian-coresets_amd")
import gpu_quick
from bayesiancoresets_amd import _native as nat
gpu_quick.run(nat.ALG_GIGA, 1000000, 256, store=nat.F64, iters=20)
gpu_quick.run(nat.ALG_FW, 1000000, 512, store=nat.F64, iters=20)
PY
done
