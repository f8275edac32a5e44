#!/bin/bash
# Dev tool: optimize() at k = 1497, d = 1024 (k > d: hundreds of pivots) over workgroup shapes of optimize_lh_kernel.
#   tools/optimize_ab.sh  -> gpurun_out/optimize_ab.txt
export BCX_DEV=1
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/optimize_ab.txt; mkdir -p gpurun_out; : > $O
for cfg in "0 0" "1024 64" "512 64" "512 128" "256 128" "256 256" "512 96"; do
  set -- $cfg
  echo "== BCX_OPT_THREADS=$1 BCX_OPT_WGS=$2 (0 = the library's choice)" >> $O
  T=$1; W=$2
  ( [ $T -gt 0 ] && export BCX_OPT_THREADS=$T; [ $W -gt 0 ] && export BCX_OPT_WGS=$W
    timeout 300 python tools/optimize_bench.py 1,1000000,1024,1500 2>&1 | grep -v amdgpu.ids >> $O )
done
echo "== the barriers' forms at the library's shape (512 x 128): fences back (BCX_GRID_FENCE=1), every workgroup polling the arrival counter (BCX_GRID_FLAT=1), both, the default again" >> $O
for env in "BCX_GRID_FENCE=1" "BCX_GRID_FLAT=1" "BCX_GRID_FENCE=1 BCX_GRID_FLAT=1" "BCX_NONE=1"; do
  echo "-- $env" >> $O
  ( export $env; timeout 300 python tools/optimize_bench.py 1,1000000,1024,1500 2>&1 | grep -v amdgpu.ids >> $O )
done
echo "== k = 999, d = 512 (GIGA)" >> $O
for cfg in "0 0" "512 64" "256 64" "512 32"; do
  set -- $cfg; T=$1; W=$2
  echo "-- BCX_OPT_THREADS=$T BCX_OPT_WGS=$W" >> $O
  ( [ $T -gt 0 ] && export BCX_OPT_THREADS=$T; [ $W -gt 0 ] && export BCX_OPT_WGS=$W
    timeout 300 python tools/optimize_bench.py 0,1000000,512,1000 2>&1 | grep -v amdgpu.ids >> $O )
done
cat $O
