#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_projection.py -q -p no:cacheprovider > $O/proj.log 2>&1; echo "proj rc=$? :: $(tail -n 1 $O/proj.log)"
timeout 600 python tools/proj_ab.py > $O/proj_ab.txt 2>&1; grep -v amdgpu $O/proj_ab.txt | tail -9
