#!/bin/bash
# Per-kernel durations of one bench run (rocprofv3 kernel trace + stats).  Run on the GPU box from the repo root:
#   bash tools/profile_kernels.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/...kernel_stats.csv
set -e
TAG=$1; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_$TAG/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), r['AverageNs'])
PY
