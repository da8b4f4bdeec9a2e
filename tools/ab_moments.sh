#!/bin/bash
# A/B of two builds of the kernel library on the moments kernel ALONE (tools/kbench.py: N = 1e6, 5 launches back to back), alternating
# on one box over (fp64, M = 1024), (fp64, M = 512), (fp32, M = 1024).  Builds: ab/lib<x>.so from `tools/ab_builds.sh build …` or hipcc -D….
#   box: bash tools/ab_moments.sh ab/libprev.so t-svgp_amd/csrc/libtsvgp_hip.so > gpurun_out/<dir>/ab_moments.txt
LIBS=("$@")
for spec in "f64 1024" "f64 512" "f32 1024"; do
  read dt M <<< "$spec"
  for rep in 1 2; do for lib in "${LIBS[@]}"; do
    echo "== $lib dtype $dt M $M (N = 1e6, 5 launches back to back)"
    TSVGP_HIP_LIB=$PWD/$lib python tools/kbench.py --rows 1000000 --M $M --dtype $dt --reps 5 2>/dev/null | grep "moments upper"
  done; done
done
