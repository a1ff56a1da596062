set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools/probes && hipcc --offload-arch=gfx950 -O3 spmm_patch_sweep.hip -o /tmp/spmm_patch_sweep 2>/dev/null
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcp_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_$c -- /tmp/spmm_patch_sweep > $R/gpurun_out/pmcp_$c.log 2>&1
  echo "pmc $c exit $?"
done
cd $R/gpurun_out
python3 - <<'PY'
import csv, glob, collections, re
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f'pmcp_{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c: continue
            k = row['Kernel_Name'].split('(')[0]
            agg[k] += float(row['Counter_Value']); n[k] += 1
    for k in agg: out.setdefault(k, {})[c] = agg[k] / n[k]; out[k]['n'] = n[k]
for k, v in out.items():
    print(f"{k:60s} x{v['n']:3d}  fetch {2 * v.get('FETCH_SIZE', 0) * 1024 / 1e6:8.1f} MB  write {v.get('WRITE_SIZE', 0) * 1024 / 1e6:8.1f} MB")
PY
find $R/gpurun_out/pmcp_* -name "*.csv" -size +8M -delete
