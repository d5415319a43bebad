#!/bin/bash
# One experimental library under several settings of one environment variable:
#   tools/exp_env.sh NAME VAR value...     (GPU box, repo root)  ->  one bench line per value
n=$1; var=$2; shift 2
for v in "$@"; do
  env $var=$v TOPO_HIP_LIB=$PWD/exp/libtopo_$n.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 > gpurun_out/exp_${n}_$v.log 2> gpurun_out/exp_${n}_$v.err || { echo "$n $var=$v FAILED"; tail -3 gpurun_out/exp_${n}_$v.err; continue; }
  python - "$n $var=$v" gpurun_out/exp_${n}_$v.log <<PY
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1].ljust(28), d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()})
PY
done
