#!/bin/bash
# Per-launch durations of one kernel under rocprofv3 (kernel trace), alternating launches listed separately:
#   tools/exp_trace_big.sh LIBNAME KERNEL   (LIBNAME "" = the in-tree library)
R=$(cd "$(dirname "$0")/.." && pwd)
n=${1:-base}; k=${2:-k_raster_big}
lib=$R/topo-renderer_amd/libtopo_hip.so; [ -n "$1" ] && lib=$R/exp/libtopo_$1.so
O=$R/gpurun_out/trace_$n; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TOPO_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $O/run.log 2>&1
python3 - "$O" "$k" <<'PY'
import sys,glob,csv,statistics
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name'] and 'overlay' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(sys.argv[2], 'launches', len(d), 'even(us)', round(statistics.median(d[0::2]),1), 'odd(us)', round(statistics.median(d[1::2]),1))
PY
