#!/bin/bash
# Runs bench.py for every named config and prints a digest of each JSON line; raw lines land in gpurun_out/bench/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/bench
for c in ${@:-c4 c2 c3 c5}; do
  echo "== $c"
  s=$(date +%s)
  python bench.py --config $c > gpurun_out/bench/$c.json 2> gpurun_out/bench/$c.err
  echo "rc=$? $(( $(date +%s) - s )) s wall"; tail -n 3 gpurun_out/bench/$c.err
  python - "$c" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/bench/%s.json" % sys.argv[1]))
except Exception as e:
    print("no JSON:", e); sys.exit(0)
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype")})
print("roofline", d["roofline"])
print("cpu_baseline", d.get("cpu_baseline"))
print("config", {k: v for k, v in d["config"].items() if k != "workload"})
PY
done
