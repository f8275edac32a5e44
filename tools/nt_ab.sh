#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
# dev: A/B the non-temporal row loads of the scan kernel
cd $GRAFT_REPO_ROOT/bayesian-coresets_amd
for rep in 1 2; do
for v in "" "-DBCX_NO_NT"; do
  touch csrc/scan.hip; make EXTRA="$v" >/dev/null 2>&1
  echo "== variant '$v'"
  python ../tools/gpu_quick.py sweep 2>&1 | grep "alg=" | cut -c1-100
done; done
touch csrc/scan.hip; make >/dev/null 2>&1
