#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s13; mkdir -p $O
timeout 900 python tests/race_hunt_omp.py 400 > $O/race_auto.txt 2>&1; tail -1 $O/race_auto.txt
BCX_OMP_THREADS=512 timeout 600 python tests/race_hunt_omp.py 200 > $O/race_512.txt 2>&1; tail -1 $O/race_512.txt
BCX_OMP_THREADS=1024 timeout 600 python tests/race_hunt_omp.py 200 > $O/race_1024.txt 2>&1; tail -1 $O/race_1024.txt
timeout 900 python tests/race_hunt.py 200 > $O/race_old.txt 2>&1; tail -6 $O/race_old.txt
