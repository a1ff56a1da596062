#!/bin/bash
# Round 2, first GPU call: the new tests, the default bench line, a kernel trace and the PMC traffic passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
echo "== new tests"
timeout 1500 python -m pytest tests/test_pipeline.py tests/test_full_size.py tests/test_module_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/a_pytest_new.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/a_pytest_new.log
echo "== bench (driver form)"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench exit $?"; cut -c1-1800 gpurun_out/a_bench.json; tail -3 gpurun_out/a_bench.err
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/a_prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/a_prof -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/a_prof.log 2>&1; echo "rocprof exit $?")
f=$(find gpurun_out/a_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/a_kernel_stats.csv && head -16 "$f" | cut -c1-200
find gpurun_out/a_prof -name "*kernel_trace.csv" -size +20M -delete
echo "== PMC traffic"
bash tools/gpu_pmc_bench.sh 2>&1 | tail -20
cp gpurun_out/spmm_traffic_bench.json gpurun_out/a_hbm_traffic_bench.json
