#!/bin/bash
# Other BASELINE.json configurations through bench.py (parity-test shapes, not the contract bench line)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { echo "### $*"; timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   value %.2f samples/s  %.1f ms/step  spmm %.0f GB/s (%.1f%%)  workload: %s' % (d['value'], d['ms_per_step'], r['achieved'], 100*r['frac'], d['config']['workload'][-95:]))
print('   top kernels:', ', '.join('%s %.1fms' % (k.replace('stc_','').replace('_f32',''), v['ms_per_step']) for k,v in list(d['kernels'].items())[:5]))
" | tee -a gpurun_out/configs.log; }
rm -f gpurun_out/configs.log
run --grid 100 --order 3 --batch-per-gpu 4            # config 4: N=10 000, C=32, T=24, K=3
run --grid 100 --order 2 --batch-per-gpu 4
run --grid 224 --order 3 --batch-per-gpu 2            # metric shape at K=3 (1 sample: K=3 saves one more slab per convolution)
run --grid 224 --categories 64 --batch-per-gpu 2      # config 5 width in fp32 (bf16 has no reference behaviour)
run --grid 224 --permute                              # random node order, renumbered internally
run --grid 224 --permute --no-reorder                 # random node order as given
run --grid 224 --batch-per-gpu 1
run --grid 224 --batch-per-gpu 2
run --grid 224 --batch-per-gpu 4
run --grid 224 --batch-per-gpu 6
