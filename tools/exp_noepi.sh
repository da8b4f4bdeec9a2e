#!/bin/bash
# Round 4 ablation: the moments kernel with its per-column-tile square-sum epilogue left out (-DTSVGP_EXP_NOEPI build at ab/libnoepi.so)
# against the product build, tools/kbench.py alternating on one box: the difference is what a second accumulator set could AT MOST hide.
#   here:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DTSVGP_EXP_NOEPI -I include t-svgp_amd/csrc/tsvgp_kernels.hip -o ab/libnoepi.so
#   box :  bash tools/exp_noepi.sh > gpurun_out/<dir>/noepi.txt
for spec in "f64 1024" "f64 512" "f32 1024"; do
  set -- $spec
  for rep in 1 2; do for lib in ab/libnoepi.so t-svgp_amd/csrc/libtsvgp_hip.so; do
    echo "== $lib dtype $1 M $2 (N = 1e6, 5 launches back to back)"
    TSVGP_HIP_LIB=$PWD/$lib python tools/kbench.py --rows 1000000 --M $2 --dtype $1 --reps 5 2>/dev/null | grep "moments upper"
  done; done
done
