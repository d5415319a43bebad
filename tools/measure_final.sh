set -e
R=$(pwd); O=$R/gpurun_out/r03g; mkdir -p $O
for w in c1 c2 c3; do timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w done"; done
timeout -k 10 900 python3 bench.py --workload c4 --check --also-pipelined > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 done"
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default done"
timeout -k 10 300 python3 bench.py --workload c4 --pipeline 2 --steps 20 --no-cpu-baseline --no-pmc > $O/bench_c4_pipelined.json 2> $O/bench_c4_pipelined.err; echo "pipelined done"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu-baseline --no-pmc --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 $R/bench.py --workload c4 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra > $O/stats.log 2>&1; echo "stats done"
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*kernel_trace.csv" -size +8M -delete
for f in c1 c2 c3 c4 default c4_pipelined c5; do python3 tools/bench_brief.py < $O/bench_$f.json | cut -c1-200; done
head -3 $O/kernel_stats.csv | cut -c1-200
