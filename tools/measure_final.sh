set -e
R=$(pwd); O=$R/gpurun_out/r02c; mkdir -p $O
timeout -k 10 900 python3 bench.py --check --also-pipelined > $O/bench_c4.json 2> $O/bench_c4.err; echo c4 done
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo default done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 $R/bench.py --no-cpu-baseline --no-pmc > $O/stats.log 2>&1; echo stats done
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep -E "k_resolve|k_raster_big" $O/kernel_stats.csv | cut -c1-160
tail -c 300 $O/bench_default.json
