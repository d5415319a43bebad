#!/bin/bash
# Builds the calibration micro-benchmarks (tools/calib.hip) -> tools/_build/calib (git-ignored; travels to the GPU box).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o $R/tools/_build/calib $R/tools/calib.hip
