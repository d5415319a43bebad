#!/bin/bash
# rocprofv3 kernel stats of bench.py for one library under one environment setting:
#   tools/exp_stats.sh NAME [VAR=value ...]      (GPU box, repo root)  ->  gpurun_out/stats_NAME*/, prints the frame kernels' rows
n=$1; shift
R=$(pwd); tag=$n$(echo "$@" | tr -c 'A-Za-z0-9' '_')
( cd /tmp && env TMPDIR=/tmp TOPO_HIP_LIB=$R/exp/libtopo_$n.so "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$tag -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $R/gpurun_out/stats_$tag.log 2>&1 )
f=$(find $R/gpurun_out/stats_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag"; python3 - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r'(k_\w+)',r['Name'])
    if m and not m.group(1).startswith(('k_normals','k_block')): print(m.group(1).ljust(22), 'calls',r['Calls'].rjust(5),'avg_us',round(float(r['AverageNs'])/1e3,1),'min',round(float(r['MinNs'])/1e3,1),'max',round(float(r['MaxNs'])/1e3,1))
PY
