#!/bin/bash
# SQ counters of the frame kernels, one --pmc pass per group (kernel trace only, as gpurun requires).
#   bash tools/pmc_kernel.sh <tag> "<CTR CTR ...>" ["<CTR ...>" ...]   -> gpurun_out/pmc_<tag>/<n>/...counter_collection.csv
set -e
TAG=$1; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
n=0
for grp in "$@"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$TAG/$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
  n=$((n+1))
done
python3 $R/tools/summarize_pmc.py $R/gpurun_out/pmc_$TAG
exit 0
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/pmc_$TAG/*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if not k.startswith('k_r') and not k.startswith('k_o') and not k.startswith('k_c'): continue
    print(k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())}, 'launches', len(next(iter(v.values()))))
PY
