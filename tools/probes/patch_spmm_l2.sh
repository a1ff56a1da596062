#!/bin/bash
# L2 (TCC) request / hit / miss / read-request counters of the aggregation kernels on the bench's unit, and of the standalone tile probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
[ -x /tmp/spmm_patch_sweep ] || (cd $R/tools/probes && hipcc --offload-arch=gfx950 -O3 spmm_patch_sweep.hip -o /tmp/spmm_patch_sweep 2>/dev/null)
i=0
for set in "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/l2a_$i $R/gpurun_out/l2b_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/l2a_$i -- python3 $R/tools/probes/patch_spmm_unit.py > $R/gpurun_out/l2a_$i.log 2>&1; echo "lib pass $i exit $?"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/l2b_$i -- /tmp/spmm_patch_sweep > $R/gpurun_out/l2b_$i.log 2>&1; echo "probe pass $i exit $?"
done
cd $R/gpurun_out
python3 - <<'PY'
import csv, glob, collections, re
out = collections.defaultdict(dict)
for f in glob.glob('l2[ab]_*/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|void ', '', row['Kernel_Name']).split('(')[0]
        agg[(k, row['Counter_Name'])] += float(row['Counter_Value']); n[(k, row['Counter_Name'])] += 1
    for (k, c), v in agg.items():
        out[k][c] = v / n[(k, c)]
keep = [k for k in out if 'spmm' in k or k.startswith('copy') or 'patch_pipe_kernel<4, 8, 256' in k or 'patch_kernel<4, 8, 128, 512' in k]
for k in keep:
    print(k[:48].ljust(48), '  '.join(f'{c.replace("TCC_", "").replace("_sum", "")}={v / 1e6:8.2f}M' for c, v in sorted(out[k].items())))
PY
find $R/gpurun_out/l2* -name "*.csv" -size +8M -delete
