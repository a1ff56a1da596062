#!/bin/bash
# Refresh what depends on the kernel sources after a late change: quick GPU suite, PMC traffic file, driver-form bench line.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/z_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/z_pytest_gpu.log
cp gpurun_out/parity_errors.txt gpurun_out/z_parity_errors.tsv 2>/dev/null
bash tools/gpu_pmc_bench.sh 2>&1 | tail -6
cp gpurun_out/spmm_traffic_bench.json gpurun_out/z_hbm_traffic_bench.json
mkdir -p profiles/r02 && cp gpurun_out/z_hbm_traffic_bench.json profiles/r02/hbm_traffic_bench.json      # so that the line below quotes it
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err; echo "bench exit $?"
python3 -c "import json;d=json.load(open('gpurun_out/z_bench.json'));r=d['roofline'];print(round(d['value'],3), round(d['ms_per_step'],2), r['traffic'], round(r['frac'],4), round(r['unit_d3']['frac'],4))"
