#!/bin/bash
# Runs bench.py once per experimental library: [EXP_ARGS="--bench-flag ..."] tools/exp_run.sh NAME... (on the GPU box, from the repo root)
for n in "$@"; do
  lib=${n%%:*}; export TOPO_REGION_GRID=${n#*:}; [ "$lib" = "$n" ] && unset TOPO_REGION_GRID; n=$lib
  TOPO_HIP_LIB=$PWD/exp/libtopo_$n.so TOPO_DEBUG_COUNTERS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 $EXP_ARGS > gpurun_out/exp_$n.log 2> gpurun_out/exp_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/exp_$n.err; continue; }
  python - "$n" <<PY
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/exp_{n}.log").read().strip().split("\n")[-1])
print(n.ljust(12), d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()})
PY
  grep "topo. counters" gpurun_out/exp_$n.err | tail -1
done
