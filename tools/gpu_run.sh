#!/bin/bash
# One parametrised GPU job script (run via gpurun):  bash tools/gpu_run.sh <step> [<step> ...]   -- steps run in order, stop at the first failure.
#   tests            the whole -m gpu suite -> gpurun_out/pytest_gpu.log (+ parity_errors.txt)
#   tests:<expr>     pytest -k <expr>
#   bench            default bench line (driver form, 20 steps) -> gpurun_out/bench_default.json
#   quick            2-step bench without the CPU baseline, per-kernel table printed
#   presets          --preset cfg2 (K = 2, 3) / cfg4 / cfg5 / sf / sf-learned lines -> gpurun_out/preset_*.json
#   stats            rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/kernel_stats.csv
#   traffic          FETCH_SIZE / WRITE_SIZE PMC passes (tools/gpu_pmc_bench.sh) -> gpurun_out/spmm_traffic_bench.json
#   mfma             SQ / GRBM PMC passes (tools/gpu_pmc_mfma.sh) -> gpurun_out/mfma_util_f32.json
# BENCH_ARGS is appended to every bench command.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out
for step in "$@"; do
  case "$step" in
    tests) rm -f gpurun_out/parity_errors.txt
           timeout -k 10 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=12 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -4 gpurun_out/pytest_gpu.log ;;
    tests:*) timeout -k 10 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "${step#tests:}" > gpurun_out/pytest_sel.log 2>&1; rc=$?; tail -15 gpurun_out/pytest_sel.log ;;
    bench) timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; rc=$?
           python -c "import json;d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1]);print('value',round(d['value'],3),'ms',round(d['ms_per_step'],2),'frac',round(d['roofline']['frac'],4),'dominant',d['roofline'].get('dominant',{}).get('frac'),'cpu',d.get('cpu_baseline',{}).get('value'))" ;;
    quick) timeout -k 10 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/bench_quick.log 2>&1; rc=$?
           python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_quick.log').read().strip().split('\n')[-1])
print('value', round(d['value'], 3), 'samples/s  ms_per_step', round(d['ms_per_step'], 2), ' spmm GB/s', round(d['roofline']['achieved'], 1))
for k, v in d['kernels'].items():
    print(f"   {k:32s} {int(v['launches']):4d} {v['ms_per_step']:8.2f} ms  {1e3 * v['ms_per_step'] * d['steps'] / v['launches']:8.1f} us/launch  {v.get('GBps', 0):7.0f} GB/s")
PY
           ;;
    presets) rc=0
           for p in "cfg2 --order 2" "cfg2 --order 3" "cfg4" "cfg5" "sf" "sf-learned"; do
             n=$(echo $p | tr -d ' -'); timeout -k 10 900 python bench.py --preset $p --steps 5 --warmup 2 > gpurun_out/preset_$n.json 2> gpurun_out/preset_$n.err || rc=1
             python -c "import json,sys;d=json.loads(open('gpurun_out/preset_$n.json').read().strip().split('\n')[-1]);print('$p',round(d['value'],2),d['unit'],round(d['ms_per_step'],3),'ms')" || rc=1
           done ;;
    stats) cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof
           timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $R/gpurun_out/prof.log 2>&1; rc=$?
           cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -12 gpurun_out/kernel_stats.csv | cut -c1-200
           find gpurun_out/prof -name "*.csv" -size +4M -delete ;;
    traffic) bash tools/gpu_pmc_bench.sh; rc=$? ;;
    mfma) bash tools/gpu_pmc_mfma.sh; rc=$? ;;
    *) echo "unknown step $step"; rc=2 ;;
  esac
  echo "[$step] exit $rc"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
