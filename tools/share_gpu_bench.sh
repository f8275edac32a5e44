#!/bin/bash
# dev: N ranks sharing cuda:0 (gloo for the host-side collectives): per-iteration cost of the two exchange modes
# usage: tools/share_gpu_bench.sh <ranks> <rows> <dim> <alg> <steps>
R=${1:-2}; N=${2:-200000}; D=${3:-512}; ALG=${4:-fw}; K=${5:-2000}
cd "$(dirname "$0")/.." || exit 1
for mode in mailbox collective; do
  echo "== $mode"
  BCX_EXCHANGE=$mode BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 500)) bench.py --gpus $R --steps $K --warmup 50 --rows $N --dim $D --alg $ALG 2>/dev/null \
    | python -c "import sys,json; [print({k:j[k] for k in ('value','ms_per_step')}, j['config']['exchange'], j['roofline']['avg_launch_ms']) for j in map(json.loads, (l for l in sys.stdin if l.startswith('{')))]"
done
echo "== single shard"
python bench.py --steps $K --warmup 50 --rows $N --dim $D --alg $ALG --no-cpu-baseline 2>/dev/null | python -c "import sys,json; [print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['avg_launch_ms']) for j in map(json.loads, (l for l in sys.stdin if l.startswith('{')))]"
