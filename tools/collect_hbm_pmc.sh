#!/bin/bash
# Collects the HBM traffic of the frame kernels exactly as MI355X_MICROARCH.md prescribes: separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel trace only.  Run on the GPU box from the repo root:
#   bash tools/collect_hbm_pmc.sh [workload]      -> gpurun_out/hbm_pmc/{FETCH_SIZE,WRITE_SIZE}/...csv
set -e
WL=${1:-c4}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/hbm_pmc/$c -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra > $R/gpurun_out/hbm_pmc_$c.log 2>&1
done
python3 $R/tools/summarize_hbm_pmc.py $R/gpurun_out/hbm_pmc $WL
