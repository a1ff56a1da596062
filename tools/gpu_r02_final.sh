#!/bin/bash
# Round 2 evidence run: GPU suite, smoke, the driver-form bench line, rocprofv3 kernel stats, PMC HBM traffic and matrix-core
# utilisation of the same command, the other BASELINE configurations, a 2-rank rehearsal of the self-launching bench.
# Everything lands in gpurun_out/z_*; what is judged is copied into profiles/r02/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
mkdir -p gpurun_out && export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.txt
echo "== rocminfo"; (rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -4) 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/z_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/z_pytest_gpu.log
cp gpurun_out/parity_errors.txt gpurun_out/z_parity_errors.tsv 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/z_smoke.log
echo "== bench (driver form)"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err; echo "bench exit $?"; cut -c1-400 gpurun_out/z_bench.json
echo "== rocprofv3 kernel trace of the same command (3 steps)"
rm -rf gpurun_out/z_prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/z_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/z_prof.log 2>&1; echo "rocprof exit $?")
f=$(find gpurun_out/z_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/z_kernel_stats.csv && head -12 "$f" | cut -c1-160
rm -rf gpurun_out/z_prof
echo "== PMC traffic"
bash tools/gpu_pmc_bench.sh 2>&1 | tail -16
cp gpurun_out/spmm_traffic_bench.json gpurun_out/z_hbm_traffic_bench.json
echo "== MFMA utilisation"
BENCH_ARGS="--no-unit-d3" TAG=z_f32_b5 bash tools/gpu_pmc_mfma.sh 2>&1 | tail -12
echo "== other configurations"
bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/z_other_configs.txt | tail -40
echo "== bf16 storage"
for c in 64 32; do
  b=5; [ $c = 32 ] && b=10
  timeout 600 python3 bench.py --storage bf16 --categories $c --batch-per-gpu $b --steps 4 --warmup 2 --no-cpu-baseline --no-unit-d3 > gpurun_out/z_bench_bf16_c$c.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/z_bench_bf16_c$c.json'));print('bf16 C=$c', round(d['value'],2), 'samples/s', round(d['ms_per_step'],1), 'ms', round(d['roofline']['aggregate']['achieved']), 'GB/s agg')"
done
echo "== SF shape"
(timeout 300 python tools/bench_sf.py --mode csr-fixed 2>&1 | tail -1; timeout 300 python tools/bench_sf.py --mode csr-fixed --graph --steps 50 2>&1 | tail -2; timeout 600 python tools/bench_sf.py --mode dense-learned --steps 5 2>&1 | tail -1) | tee gpurun_out/z_sf_shape.txt
echo "== 2-rank rehearsal of python bench.py --gpus 2 (gloo, both ranks on this one GPU; NOT an RCCL number)"
STC_DIST_BACKEND=gloo STC_DIST_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch-per-gpu 2 --no-unit-d3 > gpurun_out/z_bench_2rank_gloo.json 2> gpurun_out/z_bench_2rank.err; echo "exit $?"
python3 -c "import json;d=json.loads([l for l in open('gpurun_out/z_bench_2rank_gloo.json') if l.startswith('{')][0]);print('2 ranks', d['n_ranks_seen'], round(d['value'],2), 'samples/s', round(d['ms_per_step'],1), 'ms', d['step_breakdown']['grad_allreduce_ms'])"
