#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "large_active_set or long_rows" > $O/parity_new.log 2>&1; echo "parity_new rc=$? :: $(tail -n 1 $O/parity_new.log)"
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/s12/bench_default.json') if l.startswith('{')][0])
print({k:j[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype','exact_mode_its','exact_mode_frac','cpu_baseline_onepass')})
print(j['roofline']); print(j['cpu_baseline'])
P
( time python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err ) 2>&1 | grep real
tail -c 600 $O/bench_c5.json
