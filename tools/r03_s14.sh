#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); O=$R/gpurun_out/s14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "c2:--config c2 --steps 400 --warmup 20" "c3:--config c3" "shard:--rows 1250304 --dim 512 --alg fw --steps 400 --warmup 20" "c4:--steps 100 --warmup 10 --no-exact-mode"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rm -rf $O/raw_$name
  BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace -d $O/raw_$name -o t -- python $R/bench.py $args --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  echo "== $name"; python $R/tools/gap_report.py $(find $O/raw_$name -name "*.db" | head -1) | tee $O/gaps_$name.txt
  rm -rf $O/raw_$name
done
