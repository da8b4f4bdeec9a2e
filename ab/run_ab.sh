#!/bin/bash
out=gpurun_out/r2w/ab.txt; mkdir -p gpurun_out/r2w; : > $out
for lib in "$@"; do TSVGP_HIP_LIB=$PWD/$lib python ab/fill_alone.py 2>/dev/null >> $out || exit 1; done
for rep in 1 2; do
  for lib in "$@"; do
    for rows in 1000000; do
      TSVGP_HIP_LIB=$PWD/$lib python bench.py --rows $rows --no-elbo-match --no-side-lines --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$lib', $rows, d['ms_per_step'], 'potrf', k['tsvgp_potrf']['avg_ms'], 'mom', k['tsvgp_moments']['avg_ms'], 'syrk', k['tsvgp_site_accum']['avg_ms'], 'fill', k['tsvgp_se_fill']['avg_ms'])" >> $out || exit 1
    done
  done
done
