#!/bin/bash
# Run on the GPU box (via gpurun): GPU test-suite, smoke, a short bench and a rocprofv3 kernel trace.
# Everything judged later is copied out of gpurun_out/ into profiles/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
echo "== rocminfo"; (rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6) 2>&1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -5 gpurun_out/bench.log
echo "== torchrun single-rank launch path + SF shape"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun.log 2>&1; echo "torchrun exit $?"; tail -1 gpurun_out/bench_torchrun.log | cut -c1-200
timeout 300 python tools/bench_sf.py --mode csr-fixed 2>&1 | tail -1 | tee gpurun_out/bench_sf.log
timeout 300 python tools/bench_sf.py --mode csr-fixed --graph --steps 50 2>&1 | tail -3 | tee -a gpurun_out/bench_sf.log
timeout 600 python tools/bench_sf.py --mode dense-learned --steps 5 2>&1 | tail -1 | tee -a gpurun_out/bench_sf.log
if [ "${PROFILE:-1}" = "1" ]; then
  echo "== rocprofv3 kernel trace"
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/prof.log 2>&1
  echo "rocprof exit $?"; tail -3 gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep only the small summaries (the raw trace can be large)
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
if [ "${KBENCH:-1}" = "1" ]; then
  echo "== per-kernel timings"
  timeout 600 python tools/bench_kernels.py > gpurun_out/kbench.log 2>&1; echo "kbench exit $?"; cat gpurun_out/kbench.log
  timeout 300 python tools/bench_kernels.py --only spmm --no-tile 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/kbench.log
  timeout 300 python tools/bench_kernels.py --only spmm --permute 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/kbench.log
  timeout 300 python tools/bench_kernels.py --only spmm --permute --no-tile 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/kbench.log
  for v in ${SPMM_VARIANTS:-}; do
    STC_SPMM_VARIANT=$v timeout 300 python tools/bench_kernels.py --only spmm 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/kbench_spmm.log
    STC_SPMM_VARIANT=$v timeout 300 python tools/bench_kernels.py --only spmm --permute 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/kbench_spmm.log
  done
fi
