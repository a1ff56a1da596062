#!/bin/bash
# PMC passes over tools/prof_node.py (counters in their own runs: no sys/hip/hsa trace domains with --pmc)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; rm -rf $R/gpurun_out/pmc_$tag; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/prof_node.py > $R/gpurun_out/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"; tail -2 $R/gpurun_out/pmc_$tag.log; }
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy)\b" | sort -u > $R/gpurun_out/counters.txt; wc -l $R/gpurun_out/counters.txt
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R/gpurun_out
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('pmc_*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0][-60:]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); 
    seen=set()
    for f in files:
        for row in csv.DictReader(open(f)):
            key=(row['Kernel_Name'], row['Dispatch_Id'])
            if key not in seen:
                seen.add(key); cnt[row['Kernel_Name'].split('(')[0][-60:]] += 1
    print('==', d)
    for k, v in agg.items():
        if 'at::' in k or 'rocclr' in k: continue
        print(f'  {k}  x{cnt[k]}: ' + '  '.join(f'{c}={val/max(1,cnt[k]):.4g}' for c, val in sorted(v.items())))
PY
