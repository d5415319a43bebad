#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the 100-tile normals launch for shapes of tools/exp_roll.py:  tools/exp_roll_pmc.sh SHAPE...   (GPU box)
R=$(pwd)
for sh in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/rollpmc_${sh}_$c -o p -- python3 $R/tools/exp_roll.py $sh > /dev/null 2>&1 )
  done
  python3 - "$sh" "$R/gpurun_out" <<'PY'
import csv,glob,sys
sh,d=sys.argv[1:]
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    best=0
    for f in glob.glob(f"{d}/rollpmc_{sh}_{c}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_normals_" in r["Kernel_Name"] and "border" not in r["Kernel_Name"] and r["Counter_Name"]==c:
                best=max(best,float(r["Counter_Value"]))
    out[c]=round(best/1024,1)
print(sh, "largest launch, MiB:", out, "(FETCH_SIZE counts half the bytes on gfx950)")
PY
done
