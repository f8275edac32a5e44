export BCX_DEV=1   # the library reads its dev switches only under this gate (csrc/dev_util.h)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in 0 1; do
rm -rf /tmp/fc$f
BCX_FLAT_WAVES_PER_CU=8 BCX_SCAN_FLAT=$f rocprofv3 --pmc FETCH_SIZE -d /tmp/fc$f -o p -- python $R/tools/shape_sweep.py 100 300 > /dev/null 2>&1
echo "FLAT=$f"; python $R/tools/rocpd_summary.py $(find /tmp/fc$f -name "*.db" | head -1) | grep -E "scan.*FETCH_SIZE" | cut -c1-60,90-200
done
