#!/bin/bash
# A/B of experimental libraries in one box: tools/exp_ab.sh NAME...  (exp/libtopo_NAME.so; 40 timed steps each, listed order, twice)
run() {
  n=$1; lib=$2
  TOPO_HIP_LIB=$PWD/exp/libtopo_$lib.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 40 > gpurun_out/ab_$n.log 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; return; }
  python3 tools/bench_brief.py $n < gpurun_out/ab_$n.log
}
for pass in 1 2; do for lib in "$@"; do run ${lib}_$pass $lib; done; done
