#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s7; mkdir -p $O
tools/run_gpu_tests.sh r03a > $O/gputests.txt 2>&1; tail -12 $O/gputests.txt
timeout 900 python tools/shape_sweep.py > $O/shape_sweep.txt 2>&1; tail -30 $O/shape_sweep.txt
