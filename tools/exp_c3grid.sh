for g in 512 1024 2048 4096 8192 0; do
  TOPO_RESOLVE_GRID=$g timeout -k 10 200 python bench.py --workload c3 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 > gpurun_out/exp_c3grid_$g.log 2> gpurun_out/exp_c3grid_$g.err; python tools/bench_brief.py "c3 grid=$g" < gpurun_out/exp_c3grid_$g.log
done
for g in 512 2048 0; do
  TOPO_RESOLVE_GRID=$g timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 10 > gpurun_out/exp_c2grid_$g.log 2> gpurun_out/exp_c2grid_$g.err; python tools/bench_brief.py "c2 grid=$g" < gpurun_out/exp_c2grid_$g.log
done
