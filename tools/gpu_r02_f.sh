#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
echo "== C=64 fp32 bench (batch 2)"
timeout 600 python3 bench.py --categories 64 --batch-per-gpu 2 --steps 4 --warmup 1 --no-cpu-baseline --no-unit-d3 > gpurun_out/f_c64_f32.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open('gpurun_out/f_c64_f32.json'))
print('C=64 f32', round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms; mem', round(d['hbm_peak_allocated_gb'],1))
for k,v in d['kernels'].items(): print('   ', k, v['launches']//4, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
echo "== MFMA utilisation C=64"
BENCH_ARGS="--categories 64 --batch-per-gpu 2 --no-unit-d3" TAG=f32_c64 bash tools/gpu_pmc_mfma.sh 2>&1 | tail -14
