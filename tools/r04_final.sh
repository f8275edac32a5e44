#!/bin/bash
# ONE final pass of round 4 at the final kernel sources: GPU suite (one process per file), smoke, the suite in one process as the
# driver runs it, then every profile part (tools/profile_round.sh r04).  Output: gpurun_out/final, gpurun_out/prof_r04;
# tools/collect_r04.sh copies the summaries into profiles/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final; mkdir -p $O
tools/run_gpu_tests.sh r04final > $O/gputests.txt 2>&1; tail -8 $O/gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gputests_single.txt 2>&1; tail -1 $O/gputests_single.txt
tools/profile_round.sh r04 c4 c4pmc c2 c3 c5 c5pmc fam opt xch shards > $O/profile.log 2>&1; tail -2 $O/profile.log
python tools/proj_bench.py linreg,5000000,301,256 logistic,2000000,300,256 poisson,2000000,301,256 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r04/proj_bench_kernel_times.txt
python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r04/optimize_times.txt
BCX_OPT_GRID=1 python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[nnls_grid.hip only] /" >> gpurun_out/prof_r04/optimize_times.txt
ls gpurun_out/prof_r04 | wc -l
