#!/bin/bash
# Dev tool: the several-iterations-per-launch form (csrc/persist.hip) against one launch per kernel, interleaved on one box.
#   tools/persist_ab.sh [reps]    -> gpurun_out/persist_ab.txt
export BCX_DEV=1
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/persist_ab.txt; mkdir -p gpurun_out; : > $O
R=${1:-2}
run() {   # label, BCX_PERSIST, bench args...
  local lab=$1 p=$2; shift 2
  BCX_PERSIST=$p timeout 300 python bench.py "$@" --no-cpu-baseline --no-side-legs --no-exact-mode --no-f16-leg 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s persist=%s  %9.1f it/s  %8.2f us/it  kernel %s avg %.2f us  iter frac %.4f' % ('$lab', '$p', d['value'], d['ms_per_step']*1e3, r.get('kernel'), r['avg_launch_ms']*1e3, r['algorithmic_bytes_per_launch']/(d['ms_per_step']*1e-3)/8e12))" >> $O 2>&1
}
for i in $(seq $R); do
  for p in 0 1; do
    run "fw 10k x 100" $p --alg fw --rows 10240 --dim 100 --steps 400 --warmup 40
    run "giga 100k x 100" $p --alg giga --rows 102400 --dim 100 --steps 400 --warmup 40
    run "fw 100k x 256" $p --alg fw --rows 102400 --dim 256 --steps 400 --warmup 40
    run "giga 500k x 100" $p --alg giga --rows 500000 --dim 100 --steps 400 --warmup 40
    run "c2 giga 1M x 256" $p --config c2 --steps 400 --warmup 40
    run "shard fw 1250304 x 512" $p --alg fw --rows 1250304 --dim 512 --steps 300 --warmup 30
    run "fw 2.5M x 512 (5 GB)" $p --alg fw --rows 2500608 --dim 512 --steps 100 --warmup 10
  done
done
cat $O
