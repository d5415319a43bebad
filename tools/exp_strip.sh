#!/bin/bash
# raster strip height (cell rows per wave) against the small configurations: bash tools/exp_strip.sh  (GPU box, repo root)
for w in c1 c2 c3 c4; do
  for st in 1 2 4; do
    TOPO_NEAR_STRIP=$st timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 20 > gpurun_out/strip_${w}_$st.json 2> gpurun_out/strip_${w}_$st.err || { echo "$w $st FAILED"; continue; }
    python3 tools/bench_brief.py "$w strip $st" < gpurun_out/strip_${w}_$st.json
  done
done
