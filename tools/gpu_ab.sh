#!/bin/bash
# A/B on ONE box (gpurun): the tree's library against the libraries under gpurun_ab/ (built from other trees / with probe flags), same bench command.
#   bash tools/gpu_ab.sh bench <name> [<name> ...]     quick bench (6 steps) per library: value + the cell kernels' per-launch times
#   bash tools/gpu_ab.sh sfl <name> [...]              --preset sf-learned per library: step time + the learned-graph gradient launches
#   bash tools/gpu_ab.sh sweep <name> [...]            the heavy-graph forward cases of tests/test_scale_sweep.py per library
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
mode=$1; shift
cp stc-gnn_amd/libstc_hip.so gpurun_out/.tree.so
trap 'cp gpurun_out/.tree.so stc-gnn_amd/libstc_hip.so' EXIT      # an interrupt or a timeout must not leave a foreign library in the tree
for name in "$@"; do
  if [ "$name" = tree ]; then cp gpurun_out/.tree.so stc-gnn_amd/libstc_hip.so
  elif ! cp "gpurun_ab/$name.so" stc-gnn_amd/libstc_hip.so; then echo "$name: no gpurun_ab/$name.so, skipped"; continue; fi
  if [ "$mode" = bench ]; then
    timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.log 2>&1 || { echo "$name: bench failed"; tail -5 gpurun_out/ab_$name.log; continue; }
    python - "$name" <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/ab_{sys.argv[1]}.log').read().strip().split('\n')[-1])
k = d['kernels']
print(f"{sys.argv[1]:8s} value {d['value']:.3f}  ms {d['ms_per_step']:.2f}  " + '  '.join(f"{n.replace('stc_', '').replace('_f32', '')} {1e3 * v['ms_per_step'] * d['steps'] / v['launches']:.1f}us" for n, v in list(k.items())[:5]))
PY
  elif [ "$mode" = sf ]; then        # the SF shape on a fixed sparse graph (bench.py --preset sf)
    timeout -k 10 600 python bench.py --preset sf --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$name.log 2>&1 || { echo "$name: bench failed"; tail -5 gpurun_out/ab_$name.log; continue; }
    python - "$name" <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/ab_{sys.argv[1]}.log').read().strip().split('\n')[-1])
k = d['kernels']
print(f"{sys.argv[1]:16s} ms {d['ms_per_step']:.3f}  " + '  '.join(f"{n.replace('stc_', '').replace('_f32', '')} {1e3 * k[n]['ms_per_step'] * d['steps'] / k[n]['launches']:.1f}us x{k[n]['launches'] / d['steps']:.0f}" for n in ('stc_cell_small_bwd_f32', 'stc_cell_small_fwd_f32') if n in k))
PY
  elif [ "$mode" = sfl ]; then       # the reference's full model at the SF shape (bench.py --preset sf-learned): step time + the learned-graph gradient launches
    timeout -k 10 600 python bench.py --preset sf-learned --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.log 2>&1 || { echo "$name: bench failed"; tail -5 gpurun_out/ab_$name.log; continue; }
    python - "$name" <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/ab_{sys.argv[1]}.log').read().strip().split('\n')[-1])
k = d['kernels']
print(f"{sys.argv[1]:16s} ms {d['ms_per_step']:.3f}  " + '  '.join(f"{n.replace('stc_', '').replace('_f32', '')} {1e3 * k[n]['ms_per_step'] * d['steps'] / k[n]['launches']:.1f}us x{k[n]['launches'] / d['steps']:.0f}" for n in ('stc_mix_dt_f32', 'stc_mix_grad_f32', 'stc_graph_grad_f32', 'stc_cell_small_bwd_f32', 'stc_cell_small_fwd_f32') if n in k))
PY
  else
    rm -f gpurun_out/parity_errors.txt
    timeout -k 10 600 python -m pytest tests/test_scale_sweep.py -m gpu -q --tb=no -p no:cacheprovider -k "(heavy-graph-zero-bias and (0.001 or 1.0-h)) or graph-x16" > gpurun_out/ab_sweep_$name.log 2>&1
    echo "== $name: $(tail -1 gpurun_out/ab_sweep_$name.log)"
    grep yhat gpurun_out/parity_errors.txt | grep -v c32k3 | awk -F'\t' '{printf "   %-58s %s %s\n", $1, $3, $5}'
  fi
done
