#!/bin/bash
# round-3 GPU session 1: world-size 4/8 shared-GPU tests, bench --gpus 8 (shared), OMP path histogram on c3
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -p no:cacheprovider > $O/sharded.log 2>&1; echo "sharded rc=$?" | tee -a $O/summary.txt
tail -5 $O/sharded.log
for n in 8 4 2; do
BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus $n --steps 100 --warmup 10 > $O/bench_c4_share$n.json 2> $O/bench_c4_share$n.err; echo "bench share$n rc=$?" | tee -a $O/summary.txt
tail -c 1500 $O/bench_c4_share$n.json
done
timeout 600 python tools/omp_hist.py --rows 1000000 --itrs 140 > $O/omp_hist_c3.txt 2>&1; echo "omp_hist rc=$?" | tee -a $O/summary.txt
tail -12 $O/omp_hist_c3.txt
timeout 600 python tools/omp_hist.py --rows 1000000 --itrs 140 --randn --quiet > $O/omp_hist_randn.txt 2>&1
tail -8 $O/omp_hist_randn.txt
