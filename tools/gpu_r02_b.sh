#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
echo "== mall probe"
timeout 300 python tools/probes/mall_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b_mall_probe.txt
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/b_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/b_pytest_gpu.log
cp gpurun_out/parity_errors.txt gpurun_out/b_parity_errors.tsv 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== other configs"
timeout 600 python3 bench.py --grid 100 --order 3 --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/b_cfg4_k3.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/b_cfg4_k3.json'));print('cfg4 K=3', d['value'], d['ms_per_step'])"
timeout 600 python3 bench.py --grid 100 --order 2 --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/b_cfg4_k2.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/b_cfg4_k2.json'));print('cfg4 K=2', d['value'], d['ms_per_step'])"
