#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s6; mkdir -p $O
run() { local name=$1; shift; timeout 1500 "$@" > $O/$name.log 2>&1; echo "$name rc=$? :: $(tail -n 1 $O/$name.log)" | tee -a $O/summary.txt; }
run parity python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x
grep -n "FAILED\|Error" $O/parity.log | head
run parity_resolve env BCX_OMP_FORCE_RESOLVE=3 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "omp or OMP or F7 or optimize or repeated or wide or incremental or reset"
run parity_t1024 env BCX_OMP_THREADS=1024 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "omp or OMP or F7 or repeated"
run omp_hist_c3 python tools/omp_hist.py --rows 1000000 --itrs 140 --quiet
tail -7 $O/omp_hist_c3.log
run c3check python tools/c3_check.py --rows 200000 --itrs 250
tail -3 $O/c3check.log
run sharded python -m pytest tests/test_gpu_sharded.py -q -p no:cacheprovider -k "omp or 2- or four_and_eight or two_shards_on_one or peer_mailbox_exchange"
run fullsize python -m pytest tests/test_gpu_fullsize.py -q -p no:cacheprovider
python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" | tee -a $O/summary.txt
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/s6/bench_c3.json') if l.startswith('{')][0])
print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['avg_launch_ms'], j['config'].get('ingest_s'), j['config'].get('projection_kernel_ms'))
P
