#!/bin/bash
# Round profiles: rocprofv3 kernel traces and PMC passes (separate runs) of the bench configs and dev drivers, from the
# binary in this tree.  Output: gpurun_out/prof_$TAG/*.txt|json (summaries to copy into profiles/).
# usage: tools/profile_round.sh TAG [part ...]      parts: c4 c4pmc c2 c3 c5 c5pmc opt probe shards
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}; shift
PARTS=${@:-c4 c4pmc c2 c3 c5 c5pmc fam opt xch shards}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STAMP="source digest $(python $R/tools/stamp.py), libbcx.so sha256 $(sha256sum $R/bayesian-coresets_amd/lib/libbcx.so | cut -c1-16), head $(cat $R/bayesian-coresets_amd/lib/HEAD.txt 2>/dev/null || echo n/a)"
summ() { # name db header-lines...
  local name=$1 db=$2; shift 2
  { echo "# $STAMP"; for l in "$@"; do echo "# $l"; done; python $R/tools/rocpd_summary.py $db; } > $O/$name.txt
}
kt() { # name cmd...
  local name=$1; shift
  rm -rf $O/raw_$name
  rocprofv3 --kernel-trace --stats -d $O/raw_$name -o t -- "$@" > $O/${name}_under_rocprof.json 2> $O/${name}.err
  summ ${name}_kernel_stats $(find $O/raw_$name -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- $*"
}
pmc() { # name counters cmd...
  local name=$1 ctr=$2; shift 2
  rm -rf $O/raw_$name
  rocprofv3 --pmc $ctr -d $O/raw_$name -o p -- "$@" > $O/${name}_under_pmc.json 2> $O/${name}.err
  summ ${name} $(find $O/raw_$name -name "*.db" | head -1) "rocprofv3 --pmc $ctr -- $*"
}
for part in $PARTS; do
case $part in
c4) kt bench_c4 python $R/bench.py ;;
c4pmc) pmc scan_c4_fetch "FETCH_SIZE" python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-exact-mode --no-f16-leg --no-side-legs
       python $R/tools/scan_traffic.py $O/scan_traffic.json fw_n10000000_d512_float32=$(find $O/raw_scan_c4_fetch -name "*.db" | head -1) ;;
shards) for n in 5000192 2500608 1250304; do
         pmc scan_shard_${n}_fetch "FETCH_SIZE" python $R/bench.py --rows $n --dim 512 --alg fw --steps 20 --warmup 2 --no-cpu-baseline
         python $R/tools/scan_traffic.py $O/scan_traffic.json fw_n${n}_d512_float32=$(find $O/raw_scan_shard_${n}_fetch -name "*.db" | head -1)
       done ;;
c2) kt bench_c2 python $R/bench.py --config c2 --steps 1000 --warmup 20
    pmc scan_c2_fetch "FETCH_SIZE" python $R/bench.py --config c2 --steps 20 --warmup 2 --no-cpu-baseline
    python $R/tools/scan_traffic.py $O/scan_traffic.json giga_n1000000_d256_float32=$(find $O/raw_scan_c2_fetch -name "*.db" | head -1) ;;
c3) kt bench_c3 python $R/bench.py --config c3
    pmc scan_c3_fetch "FETCH_SIZE" python $R/bench.py --config c3 --steps 20 --warmup 2 --no-cpu-baseline
    python $R/tools/scan_traffic.py $O/scan_traffic.json omp_n1000000_d512_float32=$(find $O/raw_scan_c3_fetch -name "*.db" | head -1) ;;
c5) kt bench_c5 python $R/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-side-legs
    kt bench_c5_moments python $R/bench.py --config c5 --colsum moments --steps 3 --warmup 1 --no-cpu-baseline
    python $R/tools/kernels_after.py $(find $O/raw_bench_c5_moments -name "*.db" | head -1) moments_kernel > $O/bench_c5_moments_kernels_after_setup.txt 2>&1
    pmc proj_c5_fetch "FETCH_SIZE" python $R/tools/proj_shape.py --mode colsum --rows 5000000 --reps 4
    python $R/tools/scan_traffic.py $O/scan_traffic.json proj_colsum_linreg_n5000000_d301_s256=$(find $O/raw_proj_c5_fetch -name "*.db" | head -1) ;;
c5pmc) pmc proj_c5shard_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" python $R/tools/proj_shape.py --mode colsum --reps 8
       pmc proj_c5shard_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" python $R/tools/proj_shape.py --mode colsum --reps 8
       pmc proj_c5shard_fetch "FETCH_SIZE" python $R/tools/proj_shape.py --mode colsum --reps 8
       pmc proj_c5shard_cache "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" python $R/tools/proj_shape.py --mode colsum --reps 8
       kt proj_c5shard python $R/tools/proj_shape.py --mode colsum --reps 30
       kt proj_c5shard_select python $R/tools/proj_shape.py --mode select --reps 30
       kt proj_c5shard_write python $R/tools/proj_shape.py --mode write --reps 30 ;;
fam) # the other two likelihood families at the shard shape (N=625k, D=300, S=256): kernel trace + MFMA-busy pass per mode
     for fam in ${FAMS:-logistic poisson}; do for mode in colsum select write; do
       kt proj_${fam}_${mode} python $R/tools/proj_shape.py --family $fam --mode $mode --dim 300 --reps 30
       pmc proj_${fam}_${mode}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" python $R/tools/proj_shape.py --family $fam --mode $mode --dim 300 --reps 8
     done; done ;;
xch) # the two exchange modes, two ranks sharing this GPU (what a 1-GPU box can run): per-iteration cost next to one shard
     bash $R/tools/share_gpu_bench.sh 2 200000 512 fw 2000 > $O/exchange_modes_2ranks.txt 2>&1
     [ -n "$XCH_ONLY_SMALL" ] || BENCH_SHARE_GPU=1 python $R/bench.py --gpus 2 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_c4_2ranks_shared_gpu.json 2> $O/bench_c4_2ranks_shared_gpu.err
     [ -n "$XCH_ONLY_SMALL" ] || BENCH_SHARE_GPU=1 python $R/bench.py --gpus 8 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c4_8ranks_shared_gpu.json 2> $O/bench_c4_8ranks_shared_gpu.err ;;
opt) kt optimize python $R/tools/optimize_bench.py
     pmc optimize_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/optimize_bench.py
     kt gram python $R/tools/gram_bench.py 400,512 999,512 1497,1024 2048,2048 4096,1024
     pmc gram_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/gram_bench.py 1497,1024 4096,1024
     pmc gram_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" python $R/tools/gram_bench.py 4096,1024
     kt optimize_k1500_d2048 python $R/tools/optimize_bench.py 1,1000000,2048,1500 ;;
probe) [ -x $R/tools/probe/mfma_f64_peak ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/probe/mfma_f64_peak $R/tools/probe/mfma_f64_peak.hip
       $R/tools/probe/mfma_f64_peak > $O/mfma_f64_probe.txt 2>&1
       pmc mfma_f64_probe_pmc "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" $R/tools/probe/mfma_f64_peak ;;
esac
done
rm -rf $O/raw_*
ls -la $O
