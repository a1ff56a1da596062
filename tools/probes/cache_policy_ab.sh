set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/s13
cp stc-gnn_amd/libstc_hip.so gpurun_out/.tree.so
trap 'cp gpurun_out/.tree.so stc-gnn_amd/libstc_hip.so' EXIT
for name in tree plain; do
  if [ "$name" = tree ]; then cp gpurun_out/.tree.so stc-gnn_amd/libstc_hip.so; else cp gpurun_ab/$name.so stc-gnn_amd/libstc_hip.so; fi
  for b in 2 4 8 16; do
    timeout -k 10 300 python bench.py --preset cfg4 --no-cpu-baseline --steps 5 --warmup 2 --batch-per-gpu $b > gpurun_out/s13/cfg4_${name}_b$b.json 2>> gpurun_out/s13/err.txt || echo "$name b$b failed"
  done
  timeout -k 10 300 python bench.py --preset sf --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/s13/sf_${name}.json 2>> gpurun_out/s13/err.txt
  timeout -k 10 300 python bench.py --preset sf-learned --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/s13/sfl_${name}.json 2>> gpurun_out/s13/err.txt
  timeout -k 10 300 python bench.py --preset cfg2 --no-cpu-baseline > gpurun_out/s13/cfg2_${name}.json 2>> gpurun_out/s13/err.txt
done
