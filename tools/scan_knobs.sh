#!/bin/bash
export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
# dev: launch-width / depth knobs of the scan kernel at lane-wasting row lengths, interleaved repetitions
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for d in ${@:-200 300}; do
for cfg in "0:-1" "256:0" "256:1" "512:0" "512:1" "768:0"; do
g=${cfg%%:*}; deep=${cfg##*:}
env=""; [ "$g" != "0" ] && env="BCX_SCAN_GRID=$g"; [ "$deep" != "-1" ] && env="$env BCX_SCAN_DEEP=$deep"
echo -n "rep$rep d=$d grid=$g deep=$deep: "; env $env python tools/shape_sweep.py $d 2>&1 | grep -o "alg=[01].*8TB/s)" | sed -E 's/alg=([01]).*\((.*% of 8TB\/s)\)/alg\1 \2/' | tr '\n' ' '; echo
done; done; done
