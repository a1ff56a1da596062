#!/bin/bash
# HBM traffic of the row-blocked and the patch kernel on the bench's aggregation unit alone (tools/probes/patch_spmm_unit.py):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, per kernel, corrected as MI355X_MICROARCH.md prescribes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcu_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcu_$c -- python3 $R/tools/probes/patch_spmm_unit.py "$@" > $R/gpurun_out/pmcu_$c.log 2>&1
  echo "pmc $c exit $?"
done
cd $R/gpurun_out
python3 - <<'PY'
import csv, glob, collections, re
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f'pmcu_{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c or 'spmm' not in row['Kernel_Name']: continue
            k = re.sub(r'\(anonymous namespace\)::|void ', '', row['Kernel_Name']).split('(')[0]
            agg[k] += float(row['Counter_Value']); n[k] += 1
    for k in agg: out.setdefault(k, {})[c] = agg[k] / n[k]; out[k]['n'] = n[k]
for k, v in out.items():
    print(f"{k:50s} x{v['n']:3d}  fetch {2 * v.get('FETCH_SIZE', 0) * 1024 / 1e6:8.1f} MB  write {v.get('WRITE_SIZE', 0) * 1024 / 1e6:8.1f} MB  per launch (mean over plain and Y0 forms)")
PY
find $R/gpurun_out/pmcu_* -name "*.csv" -size +8M -delete
