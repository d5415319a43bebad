run() { # name lib envs...
  n=$1; lib=$2; shift 2
  env "$@" TOPO_HIP_LIB=$PWD/exp/libtopo_$lib.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 40 > gpurun_out/ab_$n.log 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; return; }
  python3 tools/bench_brief.py $n < gpurun_out/ab_$n.log
}
run cur1 cur X=1
run spec1 specfract X=1
run cur2 cur X=1
run spec2 specfract X=1
run wgs3 wgs3 X=1
for g in 1024 2048 8192 16384 0; do run grid$g cur TOPO_RESOLVE_GRID=$g; done
