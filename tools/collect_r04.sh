#!/bin/bash
# Copies the outputs of tools/r04_final.sh (gpurun_out/prof_r04, gpurun_out/final) into profiles/ as the tracked r04_* files.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/prof_r04
D=$(python tools/stamp.py); H=$(cat bayesian-coresets_amd/lib/HEAD.txt 2>/dev/null)
S=$(python -c "import json;print(json.load(open('$O/scan_traffic.json')).get('_stamp'))")
[ "$S" = "$D" ] || { echo "scan_traffic.json is stamped $S, the tree is $D: run tools/r04_final.sh on this tree first"; exit 1; }
for f in $(ls $O | grep -v "\.err$"); do
  case $f in
    scan_traffic.json) cp $O/$f profiles/scan_traffic.json;;
    proj_bench_kernel_times.txt|optimize_times.txt|exchange_modes_2ranks.txt|gram_times.txt|upload_rate.txt|scan_row_lengths.txt|omp_hist_c3.txt|tail_phases.txt)
      { echo "# source digest $D, head $H"; cat $O/$f; } > profiles/r04_$f;;
    *) cp $O/$f profiles/r04_$f;;
  esac
done
{ echo "# tools/run_gpu_tests.sh at the final kernel sources of round 4 (one pytest process per file, fresh MI355X box); then __graft_entry__.smoke(); then the suite in ONE process as the driver runs it"
  cat gpurun_out/final/gputests.txt; grep -v amdgpu gpurun_out/final/smoke.txt
  echo "python -m pytest tests/ -x -q -m gpu: $(tail -1 gpurun_out/final/gputests_single.txt)"; } > profiles/r04_gputests_summary.txt
echo "profiles/ refreshed from the pass at source digest $D, head $H"
