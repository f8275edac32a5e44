#!/bin/bash
# Copies the outputs of tools/r05_final.sh (gpurun_out/prof_r05, gpurun_out/final) into profiles/ as the tracked r05_* files.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/prof_r05
D=$(python tools/stamp.py); H=$(cat bayesian-coresets_amd/lib/HEAD.txt 2>/dev/null)
S=$(python -c "import json;print(json.load(open('$O/scan_traffic.json')).get('_stamp'))")
[ "$S" = "$D" ] || { echo "scan_traffic.json is stamped $S, the tree is $D: run tools/r05_final.sh on this tree first"; exit 1; }
for f in $(ls $O | grep -v "\.err$" | grep -v "_under_pmc.json$" | grep -v "omp_hist_c3\|tail_phases\|proj_c5shard_lds\|proj_c5shard_cache\|proj_c5shard_fetch\|_mfma_under\|bench_c4_8ranks"); do
  case $f in
    scan_traffic.json) cp $O/$f profiles/scan_traffic.json;;
    proj_bench_kernel_times.txt|optimize_times.txt|exchange_modes_2ranks.txt|gram_times.txt|upload_rate.txt|scan_row_lengths.txt|omp_hist_c3.txt|tail_phases.txt|svi_adam_step_pieces.txt)
      { echo "# source digest $D, head $H"; cat $O/$f; } > profiles/r05_$f;;
    *) cp $O/$f profiles/r05_$f;;
  esac
done
{ echo "# tools/run_gpu_tests.sh at the final kernel sources of round 5 (one pytest process per file, fresh MI355X box); then __graft_entry__.smoke(); then the suite in ONE process as the driver runs it"
  cat gpurun_out/final/gputests.txt; grep -v amdgpu gpurun_out/final/smoke.txt
  echo "python -m pytest tests/ -x -q -m gpu: $(tail -1 gpurun_out/final/gputests_single.txt)"; } > profiles/r05_gputests_summary.txt
python - <<'PY'
import re, os
O = 'gpurun_out/prof_r05'
D = os.popen('python tools/stamp.py').read().strip()
def parse(f):
    kern, ctr = {}, {}
    for l in open(f).read().splitlines():
        m = re.match(r'void proj_kernel<(\d), (\d), (\w+), (\d)>\(ProjArgs\)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
        if m and 'avg' not in kern:
            kern = {'inst': '<%s,%s,%s,%s>' % m.groups()[:4], 'avg': float(m.group(7)), 'min': float(m.group(8))}
        m = re.match(r'void proj_kernel<.*?\(ProjArgs\)\s+(\w+)\s+(\d+)\s+([\d.]+)', l)
        if m:
            ctr[m.group(1)] = float(m.group(3))
    return kern, ctr
rows = ["# fused projection kernel at the per-GPU shard shape of configs[4] (N=625000, S=256; D=300 for logistic / Poisson, 301 for linreg), source digest %s:" % D,
        "# average kernel duration from the rocprofv3 kernel traces (tools/proj_shape.py --reps 30), MFMA / VALU busy from separate --pmc passes",
        "# (SQ_VALU_MFMA_BUSY_CYCLES / (32 x GRBM_GUI_ACTIVE); SQ_ACTIVE_INST_VALU counts quad-cycles: x 4).  Source files: r05_proj_*_kernel_stats.txt, r05_proj_*_mfma.txt"]
for fam, Dm in (('logistic', 300), ('poisson', 300), ('c5shard', 301)):
    for mode in ('colsum', 'select', 'write'):
        name = 'proj_%s_%s' % (fam, mode) if fam != 'c5shard' else ('proj_c5shard' if mode == 'colsum' else 'proj_c5shard_' + mode)
        k, _ = parse('%s/%s_kernel_stats.txt' % (O, name))
        c = parse('%s/%s_mfma.txt' % (O, name))[1] if os.path.exists('%s/%s_mfma.txt' % (O, name)) else {}
        fl = 2.0 * 625000 * Dm * 256
        line = "%-8s %-6s %-12s avg %8.1f us = %5.1f TFLOP/s (best %7.1f us = %5.1f)" % ('linreg' if fam == 'c5shard' else fam, mode, k['inst'], k['avg'], fl / k['avg'] / 1e6, k['min'], fl / k['min'] / 1e6)
        if c.get('GRBM_GUI_ACTIVE'):
            gui = c['GRBM_GUI_ACTIVE']
            line += "  | MFMA busy %4.1f %% of SIMD cycles" % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * gui))
            if 'SQ_ACTIVE_INST_VALU' in c:
                line += ", VALU busy %4.1f %%" % (100 * 4 * c['SQ_ACTIVE_INST_VALU'] / (32 * gui))
        rows.append(line)
open('profiles/r05_proj_families_summary.txt', 'w').write("\n".join(rows) + "\n")
print("\n".join(rows[3:]))
PY
echo "profiles/ refreshed from the pass at source digest $D, head $H"
