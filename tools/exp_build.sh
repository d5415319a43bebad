#!/bin/bash
# Builds an experimental variant of the HIP library: tools/exp_build.sh NAME "-DFLAG ..." -> exp/libtopo_NAME.so
# (select it with TOPO_HIP_LIB=exp/libtopo_NAME.so).  exp/ is git-ignored but travels to the GPU box.
# Every variant leaves its recipe behind -- exp/NAME.recipe: the flags, the commit and the uncommitted diff it was built
# from -- so that a variant can be reconstructed after the fact (round 1 could not say what "exp_direct" had been).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/exp
{ echo "name: $1"; echo "flags: $2"; echo "commit: $(git -C $R rev-parse HEAD)"; echo "date: $(date -u +%FT%TZ)"; echo "--- uncommitted diff of topo-renderer_amd/csrc ---"; git -C $R diff HEAD -- topo-renderer_amd/csrc; } > $R/exp/$1.recipe
cd $R/topo-renderer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed $2 \
  -shared -o $R/exp/libtopo_$1.so topo_kernels.hip -x hip terrain_renderer.cpp topo_capi.cpp geotiff.cpp panorama.cpp -lz -ldl
