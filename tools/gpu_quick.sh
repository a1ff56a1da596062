#!/bin/bash
# GPU parity suite + one default bench line (run via gpurun); per-kernel table printed from the bench JSON.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench_quick.log 2>&1
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_quick.log').read().strip().split('\n')[-1])
print('value', round(d['value'], 3), 'samples/s  ms_per_step', round(d['ms_per_step'], 2), ' spmm GB/s', round(d['roofline']['achieved'], 1))
for k, v in d['kernels'].items():
    print(f"   {k:32s} {v['launches']:4d} {v['ms_per_step']:8.2f} ms  {1e3 * v['ms_per_step'] * d['steps'] / v['launches']:8.1f} us/launch")
PY
