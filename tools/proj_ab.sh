#!/bin/bash
# dev: A/B of two builds of the projection kernel at the configs[4] shard shape: lib/ against lib_oldproj/ (built by hand from
# an older csrc/proj.hip).  usage (GPU box): tools/proj_ab.sh
cd "$(dirname "$0")/.." || exit 1
P=bayesian-coresets_amd
run() { for fam in linreg logistic poisson; do for mode in colsum select write; do
  d=300; [ $fam = linreg ] && d=301
  python tools/proj_shape.py --family $fam --mode $mode --dim $d --reps 30 2>/dev/null | sed "s/^/$1 /" | sed 's/(all kernels of the call)//' | cut -c1-150
done; done; }
run new; run new2
mv $P/lib $P/lib_new && mv $P/lib_oldproj $P/lib
run old; run old2
mv $P/lib $P/lib_oldproj && mv $P/lib_new $P/lib
