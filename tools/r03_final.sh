#!/bin/bash
# final pass of the round at the final kernel sources: GPU suite (one process per file), smoke, every profile part
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final; mkdir -p $O
tools/run_gpu_tests.sh r03final > $O/gputests.txt 2>&1; tail -8 $O/gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
# the suite as the driver runs it: one process
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gputests_single.txt 2>&1; tail -1 $O/gputests_single.txt
tools/profile_round.sh r03 c4 c4pmc c2 c3 c5 c5pmc fam shards > $O/profile.log 2>&1; tail -2 $O/profile.log
# interleaved A/B of the projection kernels against round 2's library (built from commit b5e863e into lib_r02/)
[ -f bayesian-coresets_amd/lib_r02/libbcx.so ] && python tools/proj_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r03/proj_ab.txt
python tools/omp_hist.py --rows 1000000 --itrs 140 > gpurun_out/prof_r03/omp_hist_c3.txt 2>&1
python tools/omp_hist.py --rows 1000000 --itrs 140 --randn --quiet > gpurun_out/prof_r03/omp_hist_randn.txt 2>&1
tail -2 gpurun_out/prof_r03/omp_hist_c3.txt
