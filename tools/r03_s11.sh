#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do for fam in logistic poisson; do for nct in 4 8; do
  echo -n "nct $nct: "; BCX_PROJ_NCT=$nct timeout 300 python tools/proj_shape.py --family $fam --mode colsum --dim 300 --reps 20 2>/dev/null | tail -1
done; done; done
