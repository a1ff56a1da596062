#!/bin/bash
# Parity + A/B timing of the split-operand bf16 MFMA node kernels against the fp32 MFMA ones (run via gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
echo "== kbench x3"; timeout 600 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids | grep -E "^#|node|gates|blend" | tee gpurun_out/kbench_x3.log
echo "== kbench fp32 mfma"; STC_DISABLE_X3=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids | grep -E "^#|node" | tee gpurun_out/kbench_fp32.log
echo "== bench x3"; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x3.log 2>&1; tail -1 gpurun_out/bench_x3.log | cut -c1-330
echo "== bench fp32 mfma"; STC_DISABLE_X3=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; tail -1 gpurun_out/bench_fp32.log | cut -c1-330
