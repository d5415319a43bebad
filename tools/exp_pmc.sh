#!/bin/bash
# SQ_INSTS_VALU and duration of k_resolve for experimental libraries: tools/exp_pmc.sh NAME...   (GPU box, repo root)
R=$(pwd)
for n in "$@"; do
  lib=$R/exp/libtopo_$n.so; [ "$n" = "base" ] && lib=$R/topo-renderer_amd/libtopo_hip.so
  ( cd /tmp && TMPDIR=/tmp TOPO_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/exp_pmc_$n -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $R/gpurun_out/exp_pmc_$n.log 2>&1 )
  python3 - "$n" "$R/gpurun_out/exp_pmc_$n" <<'PY'
import csv,glob,sys,collections
n,d=sys.argv[1:]
acc=collections.defaultdict(list)
for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_resolve" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); acc["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print(n.ljust(10), {k: round(sorted(v)[len(v)//2]/ (1 if k=="dur_us" else 1e6),2) for k,v in sorted(acc.items())}, "(counters in millions)")
PY
done
