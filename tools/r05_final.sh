#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
# ONE final pass of round 5 at the final kernel sources: GPU suite (one process per file), smoke, the suite in one process as the
# driver runs it, then every profile part (tools/profile_round.sh r05).  Output: gpurun_out/final, gpurun_out/prof_r05;
# tools/collect_r05.sh copies the summaries into profiles/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final; mkdir -p $O
tools/run_gpu_tests.sh r05final > $O/gputests.txt 2>&1; tail -8 $O/gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gputests_single.txt 2>&1; tail -1 $O/gputests_single.txt
FAMS='logistic poisson' tools/profile_round.sh r05 c4 c4pmc c2 c3 c5 c5pmc fam opt xch shards > $O/profile.log 2>&1; tail -2 $O/profile.log
python tools/proj_bench.py linreg,5000000,301,256 logistic,2000000,300,256 poisson,2000000,301,256 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/proj_bench_kernel_times.txt
python tools/gram_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/gram_times.txt
python tools/upload_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/upload_rate.txt
for cfg in "fw 8192" "giga 8192" "fw 20000" "omp 16384" "fw 17" "giga 33"; do set -- $cfg; python bench.py --rows $([ $2 -gt 1000 ] && echo 150000 || echo 20000000) --dim $2 --alg $1 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 d=$2 N=%d: %.1f it/s, scan %.4f ms = %.3f of the HBM peak' % (d['config']['rows'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; done > gpurun_out/prof_r05/scan_row_lengths.txt
if [ -f bayesian-coresets_amd/lib_timing/libbcx.so ]; then   # tools/build_timing.sh (in-kernel time stamps) at the same sources
  python tools/omp_hist.py --rows 1000000 --itrs 140 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/omp_hist_c3.txt
  python tools/tail_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/tail_phases.txt
fi
python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/optimize_times.txt
BCX_OPT_COLD=1 python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[BCX_OPT_COLD=1: from the empty passive set] /" >> gpurun_out/prof_r05/optimize_times.txt
BCX_OPT_GRID=1 python tools/optimize_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[nnls_grid.hip only] /" >> gpurun_out/prof_r05/optimize_times.txt
BCX_GRAM_NTB=-1 python tools/gram_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[BCX_GRAM_NTB=-1: gram_tile_kernel of rounds 3-4] /" >> gpurun_out/prof_r05/gram_times.txt
{ python tools/svi_step_bench.py 4; BCX_SVI_DBG=9 BCX_MQ_DBG=1 python tools/svi_step_bench.py 4 | tail -9; } 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r05/svi_adam_step_pieces.txt
ls gpurun_out/prof_r05 | wc -l
