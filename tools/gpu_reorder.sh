#!/bin/bash
# A/B of the internal RCM renumbering on the randomly permuted 224x224 queen grid (run via gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for args in "" "--permute --no-reorder" "--permute"; do
  echo "== bench $args"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(json.dumps(dict(value=d['value'], ms_per_step=d['ms_per_step'], spmm_GBps=r['achieved'], spmm_us=r['avg_launch_us'], workload=d['config']['workload'])))" | tee -a gpurun_out/reorder_ab.txt
done
