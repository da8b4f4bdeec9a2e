#!/bin/bash
# A/B of alternative builds of the kernel library on ONE box (box-to-box spread is larger than most effects).
#   here:        tools/ab_builds.sh build old=<git-rev> nt="-DTSVGP_FILL_NT" ...   -> ab/lib<name>.so
#                (name=<git-rev> builds that revision's sources AGAINST THAT REVISION'S header, name="-D..." the working tree
#                with extra flags; a revision whose TSVGP_ABI_VERSION differs from the working tree's is refused at load time
#                by t-svgp_amd/_backend.py -- check out that revision's Python side to run it)
#   on the box:  gpurun -- 'bash tools/ab_builds.sh run 1000000 125000 -- ab/libold.so t-svgp_amd/csrc/libtsvgp_hip.so'
#                -> gpurun_out/ab/ab.txt: the K(X,Z) fill alone per build, then bench lines alternating over the builds
#                (REPS rounds, default 2; STEPS timed steps per line, default 20)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p ab
  for spec in "$@"; do
    name=${spec%%=*}; what=${spec#*=}
    src=t-svgp_amd/csrc/tsvgp_kernels.hip; flags=""
    inc=include
    if [ "${what#-}" = "$what" ]; then
      git show "$what:$src" > ab/_src_$name.hip; src=ab/_src_$name.hip
      mkdir -p ab/_inc_$name; git show "$what:include/tsvgp_hip.h" > ab/_inc_$name/tsvgp_hip.h; inc=ab/_inc_$name
    else flags=$what; fi
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC $flags -I $inc $src -o ab/lib$name.so
    echo "ab/lib$name.so  <-  $what"
  done
elif [ "$mode" = run ]; then
  rows=(); while [ "$1" != "--" ]; do rows+=("$1"); shift; done; shift
  out=gpurun_out/ab/ab.txt; mkdir -p gpurun_out/ab; : > $out
  for lib in "$@"; do TSVGP_HIP_LIB=$PWD/$lib python tools/fill_alone.py 2>/dev/null >> $out; done
  for rep in $(seq 1 ${REPS:-2}); do for lib in "$@"; do for n in "${rows[@]}"; do
    TSVGP_HIP_LIB=$PWD/$lib python bench.py --rows $n --steps ${STEPS:-20} --no-elbo-match --no-side-lines --cpu-budget 0 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels']
print('$lib', $n, 'ms/step', d['ms_per_step'], *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill')) if b in k])" >> $out
  done; done; done
  cat $out
else
  echo "usage: see the header of $0"; exit 2
fi
