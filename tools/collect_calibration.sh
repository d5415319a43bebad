#!/bin/bash
# Calibration of the counters and rates bench.py's roofline blocks rest on (run on the GPU box from the repo root):
#   bash tools/collect_calibration.sh  -> gpurun_out/calib/{valu.json,FETCH_SIZE/,WRITE_SIZE/} + gpurun_out/calibration.json
# (copy the latter to profiles/r02_calibration.json).  PMC passes: kernel trace only, one counter per pass, the
# program itself after `--` (MI355X_MICROARCH.md, HBM section).
set -e
R=$(pwd)
O=$R/gpurun_out/calib
mkdir -p $O
$R/tools/_build/calib valu > $O/valu.json
$R/tools/_build/calib atomics > $O/atomics.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o cal -- $R/tools/_build/calib hbm > $O/hbm_$c.log 2>&1
done
cd $R
python3 tools/summarize_calibration.py $O > gpurun_out/calibration.json
cat gpurun_out/calibration.json
