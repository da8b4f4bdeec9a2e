#!/bin/bash
# Round 4 A/B: the moments kernel's per-tile row sums by DPP rotations (product) against round 3's __shfl_xor butterfly
# (-DTSVGP_EPI_SHFL build at ab/libshfl.so), tools/kbench.py alternating on one box.   box: bash tools/exp_epi_dpp.sh
for spec in "f64 1024" "f64 512" "f32 1024"; do
  set -- $spec
  for rep in 1 2; do for lib in ab/libshfl.so t-svgp_amd/csrc/libtsvgp_hip.so; do
    echo "== $lib dtype $1 M $2 (N = 1e6, 5 launches back to back)"
    TSVGP_HIP_LIB=$PWD/$lib python tools/kbench.py --rows 1000000 --M $2 --dtype $1 --reps 5 2>/dev/null | grep "moments upper"
  done; done
done
