#!/bin/bash
# Builds an experimental variant of the HIP library: tools/exp_build.sh NAME "-DFLAG ..." -> exp/libtopo_NAME.so
# (select it with TOPO_HIP_LIB=exp/libtopo_NAME.so).  exp/ is git-ignored but travels to the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/exp
cd $R/topo-renderer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed $2 \
  -shared -o $R/exp/libtopo_$1.so topo_kernels.hip -x hip terrain_renderer.cpp topo_capi.cpp geotiff.cpp -lz
