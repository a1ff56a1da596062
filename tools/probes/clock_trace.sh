#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi, twice a second) while the default bench step runs: is the step power- or clock-limited?
#   bash tools/probes/clock_trace.sh   -> gpurun_out/clock_trace.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-unit-d3 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
: > gpurun_out/clock_trace.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power\|junction\|memory)" | tr -s ' ' | tr '\n' '|' >> gpurun_out/clock_trace.txt
  echo >> gpurun_out/clock_trace.txt
  sleep 0.5
done
wait $pid
tail -c 300 gpurun_out/clock_bench.json | head -c 200; echo
awk 'NR % 4 == 0' gpurun_out/clock_trace.txt | cut -c1-400 | tail -25
