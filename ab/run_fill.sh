#!/bin/bash
out=gpurun_out/r2w/fill.txt; mkdir -p gpurun_out/r2w; : > $out
for lib in "$@"; do TSVGP_HIP_LIB=$PWD/$lib python ab/fill_alone.py 2>/dev/null >> $out || exit 1; done
