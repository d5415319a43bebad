#!/bin/bash
# The measurements a round's profiles/ entries come from, in one GPU-box call (run from the repo root):
#   bash tools/measure_round.sh <tag>      -> gpurun_out/<tag>/...
# bench lines for c1..c5 (c4 = the default command: with its own PMC child passes, the all-cores CPU baseline and the
# whole-panorama oracle check), rocprofv3 kernel stats of the c4 command, the SQ counter passes, the LDS tile-size sweep of
# the normals kernel under rocprofv3, the counter / issue-rate calibration.
set -e
TAG=$1
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
for w in c1 c2 c3; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w done"
done
timeout -k 10 900 python3 bench.py --workload c4 --check --also-pipelined > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 done"
timeout -k 10 300 python3 bench.py --workload c4 --pipeline 2 --steps 20 --no-cpu-baseline --no-pmc > $O/bench_c4_pipelined.json 2> $O/bench_c4_pipelined.err; echo "c4 pipelined done"
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default done"
TOPO_LOAD_FUSED=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pmc --steps 5 > $O/bench_c4_separate_load.json 2> $O/bench_c4_separate_load.err; echo "separate-load done"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu-baseline --no-pmc --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 $R/bench.py --workload c4 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra > $O/stats.log 2>&1; echo "stats done"
TOPO_SWEEP_DEG=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_stats -o sweep -- python3 $R/tools/sweep_normals_lds.py > $O/normals_lds_sweep.json 2> $O/sweep.err; echo "sweep done"
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/sweep_stats -name "*kernel_stats.csv" -exec cp {} $O/normals_lds_sweep_kernel_stats.csv \;
bash tools/pmc_kernel.sh $TAG "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" > $O/pmc_summary.txt 2>&1 || true
python3 tools/summarize_pmc.py gpurun_out/pmc_$TAG --json > $O/sq_summary.json; echo "pmc done"
bash tools/collect_calibration.sh > $O/calibration.log 2>&1 && cp gpurun_out/calibration.json $O/calibration.json; echo "calibration done"
tail -c 400 $O/bench_c4.json
