#!/bin/bash
# Repeats the driver's exact GPU-suite command N times (one process each) to hunt intermittent aborts.
# usage: tools/loop_gpu_suite.sh N tag
cd "$(dirname "$0")/.." || exit 1
N=${1:-3}; TAG=${2:-loop}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  rm -f gpurun_out/current_test.txt
  [ "$i" = 1 ] && (sync; echo 3 > /proc/sys/vm/drop_caches) 2>/dev/null
  s=$(date +%s)
  timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_$i.log 2>&1
  rc=$?
  e=$(date +%s)
  echo "run $i rc=$rc $((e-s))s :: $(tail -n 1 gpurun_out/${TAG}_$i.log | cut -c1-200)" | tee -a gpurun_out/${TAG}_summary.txt
  if [ $rc -ne 0 ]; then cp gpurun_out/current_test.txt gpurun_out/${TAG}_${i}_markers.txt; tail -n 3 gpurun_out/current_test.txt | tee -a gpurun_out/${TAG}_summary.txt; fi
done
