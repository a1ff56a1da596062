#!/bin/bash
# Matrix-core utilisation of the kernels of ONE bench step from rocprofv3 --pmc passes (SQ counters and GRBM in separate runs, no
# trace domains besides --kernel-trace): per kernel  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).
# BENCH_ARGS selects the configuration; TAG names the output (gpurun_out/mfma_util_$TAG.json).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-f32}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rm -rf $R/gpurun_out/pmcm_$n; timeout 1200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $R/gpurun_out/pmcm_$n.log 2>&1; echo "pmc $n exit $?"; }
run sq SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES
run grbm GRBM_GUI_ACTIVE
cd $R/gpurun_out
BENCH_ARGS="${BENCH_ARGS:-}" python3 - "$TAG" <<'PY'
import csv, glob, json, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in ('pmcm_sq', 'pmcm_grbm'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.split(r'\(', row['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''), 1)[0]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); n[k][row['Counter_Name']] += 1
out = {}
for k, v in agg.items():
    if 'GRBM_GUI_ACTIVE' not in v or 'SQ_INSTS_MFMA' not in v or v['SQ_INSTS_MFMA'] == 0: continue
    per = {c: v[c] / n[k][c] for c in v}
    cyc = per['GRBM_GUI_ACTIVE'] / 8.0                       # counter is summed over the 8 XCDs
    out[k] = dict(launches=n[k]['GRBM_GUI_ACTIVE'], gpu_cycles_per_launch=cyc, mfma_insts_per_launch=per['SQ_INSTS_MFMA'],
                  mfma_busy_cycles_per_launch=per['SQ_VALU_MFMA_BUSY_CYCLES'], mfma_util=per['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024),
                  valu_insts_per_launch=per['SQ_INSTS_VALU'], valu_busy=4 * per['SQ_ACTIVE_INST_VALU'] / (cyc * 1024),
                  wave_cycles_stalled_on_issue=per['SQ_WAIT_INST_ANY'] / max(per['SQ_WAVE_CYCLES'], 1), wave_cycles_parked=per['SQ_WAIT_ANY'] / max(per['SQ_WAVE_CYCLES'], 1))
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
a = bench.parse(os.environ.get('BENCH_ARGS', '').split())
config_key = f'{a.storage}:{a.grid}:{a.categories}:{a.hidden}:{a.batch_per_gpu}:{a.order}:{a.layers}:{a.obs}:{a.pred}:{int(a.permute)}'
json.dump(dict(csrc_sha=bench.csrc_sha(), config_key=config_key,
               command='python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras ' + os.environ.get('BENCH_ARGS', ''),
               definition='mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); valu_busy = 4 SQ_ACTIVE_INST_VALU / the same; '
                          'SQ and GRBM counters in separate rocprofv3 --pmc passes over one step of the command',
               kernels=out), open(f'mfma_util_{sys.argv[1]}.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['gpu_cycles_per_launch'] * kv[1]['launches']):
    print(f"{k[:58]:58s} x{v['launches']:4d}  mfma_util {100 * v['mfma_util']:5.1f} %  valu_busy {100 * v['valu_busy']:5.1f} %  issue-stall {100 * v['wave_cycles_stalled_on_issue']:4.1f} %  parked {100 * v['wave_cycles_parked']:4.1f} %")
PY
find $R/gpurun_out/pmcm_* -name "*.csv" -size +8M -delete
