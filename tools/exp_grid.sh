#!/bin/bash
# bench.py with different k_resolve grid sizes (TOPO_RESOLVE_GRID: 0 = one block per workgroup): tools/exp_grid.sh [lib] G...
LIB=$1; shift
for g in "$@"; do
  TOPO_RESOLVE_GRID=$g ${LIB:+TOPO_HIP_LIB=$PWD/exp/libtopo_$LIB.so} timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 > gpurun_out/exp_grid_$g.log 2> gpurun_out/exp_grid_$g.err || { echo "grid $g FAILED"; tail -3 gpurun_out/exp_grid_$g.err; continue; }
  python tools/bench_brief.py "grid=$g" < gpurun_out/exp_grid_$g.log
done
