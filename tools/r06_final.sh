#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
# ONE final pass of round 6 at the final kernel sources: GPU suite (one process per file), smoke, the suite in one process as the
# driver runs it, the default bench line, then the profile parts.  Output: gpurun_out/final, gpurun_out/prof_r06;
# tools/collect_r06.sh copies the summaries into profiles/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final; mkdir -p $O gpurun_out/prof_r06
tools/run_gpu_tests.sh r06final > $O/gputests.txt 2>&1; tail -12 $O/gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
timeout 1700 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gputests_single.txt 2>&1; tail -1 $O/gputests_single.txt
tools/profile_round.sh r06 c4 c4pmc c2 c3 c5 c5pmc shards > $O/profile.log 2>&1; tail -2 $O/profile.log
P=gpurun_out/prof_r06
for k in 8 32 64 128 300; do tools/prof_ksweep.sh r06 $k > /dev/null 2>&1; cp gpurun_out/prof_ks_r06_k$k.txt $P/c5_adam_k$k.txt; done
BCX_LRP_DBG=1 python tools/lrp_timeline.py 2>&1 | grep -v amdgpu.ids > $P/lrp_chol_timeline.txt
tools/probe/f64_chain_probe 2>&1 | grep -v amdgpu.ids > $P/f64_chain_probe.txt
tools/probe/xcd_handoff_probe 4000 2>&1 | grep -v amdgpu.ids > $P/xcd_handoff_probe.txt
python tools/c5_ksweep.py --ks 4,8,16,24,25,32,64,128,300 --select 2>/dev/null | tail -1 > $P/c5_ksweep.json
python tools/svi_laplace_bench.py 2>/dev/null | tail -1 > $P/svi_laplace_bench.txt
python tools/svi_laplace_bench.py --family poisson --steps 4 2>/dev/null | tail -1 >> $P/svi_laplace_bench.txt
{ python tests/race_hunt_lrpost.py 2000 301 300; python tests/race_hunt_lrpost.py 400 1000 1000; python tests/race_hunt_lrpost.py 600 97 40; } 2>&1 | grep "^lrpost" > $P/race_hunt_lrpost.txt
python tools/gram_bench.py 2>&1 | grep -v amdgpu.ids > $P/gram_times.txt
python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids > $P/optimize_times.txt
{ timeout 400 python tests/race_hunt_warm.py 80; timeout 300 python tests/race_hunt.py 80; } 2>&1 | grep -v amdgpu > $P/race_hunt_optimize.txt
tools/optimize_ab.sh > /dev/null 2>&1; cp gpurun_out/optimize_ab.txt $P/optimize_ab.txt
tools/persist_ab.sh 1 > /dev/null 2>&1; cp gpurun_out/persist_ab.txt $P/persist_ab.txt
ls $P | wc -l
