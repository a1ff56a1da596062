#!/bin/bash
# bf16-storage kernels (configuration 5): parity tests + cold per-kernel timings at N = 50 176, C = 64 (run via gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_kernels.py -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:--x} > gpurun_out/pytest_bf16.log 2>&1
echo "pytest exit $?"; tail -${TAIL:-25} gpurun_out/pytest_bf16.log | cut -c1-400
if [ "${KBENCH:-1}" = "1" ]; then
timeout 600 python tools/bench_kernels.py --C 64 --only spmm-bf16 2>&1 | tee gpurun_out/kbench_bf16.txt
timeout 600 python tools/bench_kernels.py --C 64 --B 2 --only spmm-bf16 2>&1 | tee -a gpurun_out/kbench_bf16.txt
fi
