#!/bin/bash
# run-time and build-time knobs of the raster phase against the current default, in one box: bash tools/exp_knobs.sh  (GPU box, repo root)
brief() { python3 tools/bench_brief.py "$1" | cut -c1-170; }
for km in 60 75 90 105 120 150; do python3 bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 20 --occlusion-split $((km*1000)) 2>/dev/null | brief "split ${km}km"; done
for st in 2 3 4; do TOPO_NEAR_STRIP=$st python3 bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 20 2>/dev/null | brief "strip $st"; done
for lib in cur i4x16 i6x32 i8x32 rw4 rw6 rw8 cur; do TOPO_HIP_LIB=$PWD/exp/libtopo_$lib.so python3 bench.py --no-cpu-baseline --no-pmc --no-host-path --no-pipelined-extra --steps 20 2>/dev/null | brief "lib $lib"; done
