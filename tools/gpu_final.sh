#!/bin/bash
# Round-end evidence run (via gpurun): default bench, a 3-samples-per-GPU variant, kernel trace, PMC traffic of the bench.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench B=6"; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch-per-gpu 6 2>&1 | tail -1 | cut -c1-260
echo "== bench default (with cpu baseline)"; timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-260
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
echo "rocprof exit $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -24
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
bash tools/gpu_pmc_bench.sh
