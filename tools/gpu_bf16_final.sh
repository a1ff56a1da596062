#!/bin/bash
# Evidence run for the bf16-storage configuration (via gpurun): bench line, rocprofv3 kernel stats, PMC HBM traffic, cold kernel timings
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--storage bf16 --categories 64"
echo "== bench $ARGS"; timeout 900 python bench.py $ARGS --no-cpu-baseline > gpurun_out/k_bench_bf16_c64.log 2>&1; tail -1 gpurun_out/k_bench_bf16_c64.log | cut -c1-400
echo "== bench --storage bf16 (C=32, 10 samples)"; timeout 900 python bench.py --storage bf16 --batch-per-gpu 10 --no-cpu-baseline > gpurun_out/k_bench_bf16_c32.log 2>&1; tail -1 gpurun_out/k_bench_bf16_c32.log | cut -c1-400
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/prof_bf16
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf16 -- python3 bench.py $ARGS --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bf16.log 2>&1
echo "rocprof exit $?"
f=$(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/k_bf16_kernel_stats.csv && cut -c1-150 "$f" | head -16
find gpurun_out/prof_bf16 -name "*kernel_trace.csv" -size +20M -delete
BENCH_ARGS="$ARGS" bash tools/gpu_pmc_bench.sh
cp gpurun_out/spmm_traffic_bench.json gpurun_out/k_hbm_traffic_bench_bf16_c64.json
timeout 600 python tools/bench_kernels.py --C 64 --only spmm-bf16 2>&1 | tee gpurun_out/k_kbench_bf16.txt
