#!/bin/bash
# Runs the -m gpu suite one test FILE per process, so a HIP-runtime abort in one file cannot erase
# the others' results.  Logs: gpurun_out/gputests/<file>.log, one-line verdicts in summary.txt.
# usage: tools/run_gpu_tests.sh [tag] [extra pytest args]
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-run}; shift
OUT=gpurun_out/gputests_$TAG
mkdir -p "$OUT"
: > "$OUT/summary.txt"
echo "head $(cat bayesian-coresets_amd/lib/HEAD.txt 2>/dev/null || echo n/a) libbcx $(sha256sum bayesian-coresets_amd/lib/libbcx.so | cut -c1-16)" >> "$OUT/summary.txt"
rc_all=0
for f in tests/test_gpu_*.py tests/test_*.py; do
  case " $seen " in *" $f "*) continue;; esac
  seen="$seen $f"
  grep -q "mark.gpu" "$f" || continue
  name=$(basename "$f" .py)
  start=$(date +%s)
  timeout 1500 python -m pytest "$f" -q -m gpu -p no:cacheprovider "$@" > "$OUT/$name.log" 2>&1
  rc=$?
  end=$(date +%s)
  echo "$name rc=$rc $((end-start))s :: $(tail -n 1 "$OUT/$name.log")" >> "$OUT/summary.txt"
  [ $rc -ne 0 ] && [ $rc -ne 5 ] && rc_all=1
done
cp gpurun_out/current_test.txt "$OUT/markers.txt" 2>/dev/null
cat "$OUT/summary.txt"
exit $rc_all
