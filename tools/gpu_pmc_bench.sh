#!/bin/bash
# HBM traffic of every kernel of ONE default bench step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes (no trace domains besides --kernel-trace), aggregated per kernel into gpurun_out/spmm_traffic_bench.json together
# with the hash of the kernel sources and the bench configuration it was collected on (bench.py quotes roofline.traffic
# from the copy committed as profiles/rNN/hbm_traffic_bench.json (the newest round) only while both still match).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcb_$c
  timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unit-d3 --no-extras ${BENCH_ARGS:-} > $R/gpurun_out/pmcb_$c.log 2>&1
  echo "pmc $c exit $?"; tail -1 $R/gpurun_out/pmcb_$c.log | cut -c1-300
done
cd $R/gpurun_out
BENCH_ARGS="${BENCH_ARGS:-}" python3 - <<'PY'
import csv, glob, json, collections, os, re, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
a = bench.parse(os.environ.get('BENCH_ARGS', '').split())
config_key = f'{a.storage}:{a.grid}:{a.categories}:{a.hidden}:{a.batch_per_gpu}:{a.order}:{a.layers}:{a.obs}:{a.pred}:{int(a.permute)}'
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f'pmcb_{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c: continue
            k = re.split(r'\(', row['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''), 1)[0]
            agg[k] += float(row['Counter_Value']); n[k] += 1
    for k in agg:
        out.setdefault(k, {})[c + '_KB_per_launch'] = agg[k] / n[k]; out[k]['launches'] = n[k]
res = {}
for k, v in out.items():
    if 'FETCH_SIZE_KB_per_launch' in v and 'WRITE_SIZE_KB_per_launch' in v:
        v['hbm_bytes_per_launch'] = (2 * v['FETCH_SIZE_KB_per_launch'] + v['WRITE_SIZE_KB_per_launch']) * 1024
        res[k] = v
json.dump(dict(csrc_sha=bench.csrc_sha(), config_key=config_key, command='python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unit-d3 --no-extras ' + os.environ.get('BENCH_ARGS', ''),
               formula='hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM)',
               kernels=res), open('spmm_traffic_bench.json', 'w'), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:14]:
    print(f"{k[:70]:70s} x{v['launches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")
PY
find $R/gpurun_out/pmcb_* -name "*.csv" -size +8M -delete
