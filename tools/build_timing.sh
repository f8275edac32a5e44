#!/bin/bash
# dev: second copy of the library with in-kernel time stamps (-DBCX_TIMING) in bayesian-coresets_amd/lib_timing/
cd "$(dirname "$0")/../bayesian-coresets_amd" || exit 1
mkdir -p build_timing lib_timing
for f in api ingest scan resolve nnls nnls_grid omp_lh proj moments; do
  /opt/rocm/bin/hipcc -DBCX_TIMING -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function -c csrc/$f.hip -o build_timing/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_timing/*.o -o lib_timing/libbcx.so && echo built lib_timing/libbcx.so
