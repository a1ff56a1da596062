#!/bin/bash
# A/B of bf16 kernel variants on the configuration-5 train step (run via gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('  value', round(d['value'], 2), 'samples/s  ms/step', round(d['ms_per_step'], 1), ' spmm GB/s', round(d['roofline']['achieved'], 1))
for k, v in list(d['kernels'].items())[:7]:
    print(f"     {k:34s} {v['ms_per_step']:8.2f} ms  {1e3 * v['ms_per_step'] * d['steps'] / v['launches']:8.1f} us/launch")
PY
}
for w in 2 1; do
  echo "== C=64 bf16, STC_BF16_BWD_WAVES=$w"
  STC_BF16_BWD_WAVES=$w timeout 600 python bench.py --storage bf16 --categories 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$w.log 2>&1; show gpurun_out/ab_$w.log
done
echo "== C=32 bf16 B=10"
timeout 600 python bench.py --storage bf16 --steps 2 --warmup 1 --no-cpu-baseline --batch-per-gpu 10 > gpurun_out/ab_c32.log 2>&1; show gpurun_out/ab_c32.log
