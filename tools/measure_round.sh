#!/bin/bash
# The measurements a round's profiles/ entries come from, in one GPU-box call (run from the repo root):
#   bash tools/measure_round.sh <tag>      -> gpurun_out/<tag>/...
# bench lines for c1..c5 (c4 with the CPU baseline and the oracle check), rocprofv3 kernel stats of the c4 command,
# and the FETCH_SIZE / WRITE_SIZE passes behind profiles/hbm_traffic.json.
set -e
TAG=$1
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
for w in c1 c2 c3; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w done"
done
timeout -k 10 600 python3 bench.py --workload c4 --check > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 done"
timeout -k 10 600 python3 bench.py --workload c4 --pipeline 2 --steps 20 --no-cpu-baseline > $O/bench_c4_pipelined.json 2> $O/bench_c4_pipelined.err; echo "c4 pipelined done"
timeout -k 10 600 python3 bench.py --workload c5 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 $R/bench.py --workload c4 --no-cpu-baseline > $O/stats.log 2>&1; echo "stats done"
cd $R
bash tools/collect_hbm_pmc.sh c4 > $O/hbm_traffic.json 2> $O/hbm.err; echo "pmc done"
cp gpurun_out/hbm_traffic.json $O/ 2>/dev/null || true
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tail -c 600 $O/bench_c4.json
