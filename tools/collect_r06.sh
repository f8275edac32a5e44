#!/bin/bash
# Copies the outputs of tools/r06_final.sh (gpurun_out/prof_r06, gpurun_out/final) into profiles/ as the tracked r06_* files.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/prof_r06
D=$(python tools/stamp.py); H=$(cat bayesian-coresets_amd/lib/HEAD.txt 2>/dev/null)
S=$(python -c "import json;print(json.load(open('$O/scan_traffic.json')).get('_stamp'))")
# COLLECT_AT_PASS=1: the tree moved on after the pass in files no measured kernel depends on (bench.py checks the per-kernel
# stamps _stamp_scan / _stamp_proj itself): label the files with the digest the pass ran at
[ -n "$COLLECT_AT_PASS" ] && D=$S
[ "$S" = "$D" ] || { echo "scan_traffic.json is stamped $S, the tree is $D: run tools/r06_final.sh on this tree first"; exit 1; }
for f in $(ls $O | grep -v "\.err$" | grep -v "_under_pmc.json$" | grep -v "proj_c5shard_lds\|proj_c5shard_cache\|proj_c5shard_fetch\|_mfma_under"); do
  case $f in
    scan_traffic.json) cp $O/$f profiles/scan_traffic.json;;
    c5_adam_k*.txt) { echo "# rocprofv3 --kernel-trace --stats -- python tools/c5_ksweep.py --ks K --reps 2 --rows 200000  (SparseVI weight optimisation of a coreset SEEDED with K points: 3 x 100 ADAM steps, D = 301, S = 256, closed-form column sums); source digest $D, head $H"; head -18 $O/$f | cut -c1-175; } > profiles/r06_$f;;
    lrp_chol_timeline.txt|f64_chain_probe.txt|xcd_handoff_probe.txt|bench_c5_moments_kernels_after_setup.txt|svi_laplace_bench.txt|race_hunt_lrpost.txt|gram_times.txt|optimize_times.txt|optimize_ab.txt|persist_ab.txt|race_hunt_optimize.txt)
      { echo "# source digest $D, head $H"; cat $O/$f; } > profiles/r06_$f;;
    *) cp $O/$f profiles/r06_$f;;
  esac
done
{ echo "# tools/run_gpu_tests.sh at the final kernel sources of round 6 (one pytest process per file, fresh MI355X box); then __graft_entry__.smoke(); then the suite in ONE process as the driver runs it"
  cat gpurun_out/final/gputests.txt; grep -v amdgpu gpurun_out/final/smoke.txt
  echo "python -m pytest tests/ -x -q -m gpu: $(tail -1 gpurun_out/final/gputests_single.txt)"; } > profiles/r06_gputests_summary.txt
echo "profiles/ refreshed from the pass at source digest $D, head $H"
