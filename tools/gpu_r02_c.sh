#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
echo "== spmm parity tests"
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "spmm or blend or sum or post" > gpurun_out/c_pytest_spmm.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/c_pytest_spmm.log
for pipe in 0 1; do
  for B in 1 5; do
    echo "== kernel bench PIPE=$pipe B=$B"
    STC_SPMM_PIPE=$pipe timeout 300 python tools/bench_kernels.py --only spmm --B $B --iters 40 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/c_kbench_spmm.txt
for pipe in 0 1; do
  STC_SPMM_PIPE=$pipe timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c_bench_pipe$pipe.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open('gpurun_out/c_bench_pipe$pipe.json'))
print('PIPE=$pipe', round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms; plain', round(d['roofline']['achieved']), 'GB/s; agg', round(d['roofline']['aggregate']['achieved']), 'd3', round(d['roofline']['unit_d3']['achieved']))
for k,v in d['kernels'].items():
    if 'spmm' in k: print('   ', k, round(v['ms_per_step'],2), round(v.get('GBps',0)))
PY
done
