#!/bin/bash
# A/B of the BCSR rows-per-workgroup variants: cold timing + FETCH_SIZE / WRITE_SIZE per launch
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in 8 4 2; do
  echo "### STC_BCSR_BLOCKS=$v"
  STC_BCSR_BLOCKS=$v timeout 300 python3 $R/tools/bench_kernels.py --only spmm 2>&1 | grep -v amdgpu.ids
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/ab_${v}_$c
    STC_BCSR_BLOCKS=$v ITERS=3 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/ab_${v}_$c -- python3 $R/tools/prof_node.py > /dev/null 2>&1
    python3 - "$R/gpurun_out/ab_${v}_$c" $c <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'spmm' in row['Kernel_Name']:
            k = row['Kernel_Name'].split('(anonymous namespace)::')[1].split('(')[0]
            agg[k] += float(row['Counter_Value']); n[k].add(row['Dispatch_Id'])
for k in agg: print(f'   {sys.argv[2]} per launch {k}: {agg[k]/len(n[k])/1024:.1f} MB (x2 for reads on gfx950)')
PY
  done
done
