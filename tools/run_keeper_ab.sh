#!/bin/bash
# Round 5: the clock keeper beside the M x M sections (TSVGP_CLOCK_KEEPER=-1) against the default (off), alternating on ONE box.
# usage (on the box): bash tools/run_keeper_ab.sh [out dir] ["workload rows steps" ...]
R=$PWD; O=${1:-gpurun_out/r5u}; shift; mkdir -p $O; : > $O/keeper_ab.txt
line() {  # workload rows steps tag
  python bench.py --workload $1 --rows $2 --steps $3 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null > $O/line_$4_$1_$2.json
  python -c "
import json, sys
d = json.loads([l for l in open('$O/line_$4_$1_$2.json').read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$4', '$1', $2, 'ms/step', d['ms_per_step'], 'hipgraph' if d.get('hipgraph') and d['hipgraph'].get('headline_mode') == 'hipGraph replay' else 'eager',
      *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill'), ('kuu', 'tsvgp_se_fill(Kuu)')) if b in k])" >> $O/keeper_ab.txt
}
if [ $# -eq 0 ]; then set -- "ns 125000 40" "c3 125000 40" "ns 1000000 20"; fi
SPECS=("$@")
for rep in 1 2; do
  for spec in "${SPECS[@]}"; do
    set -- $spec
    line $1 $2 $3 off
    TSVGP_CLOCK_KEEPER=-1 line $1 $2 $3 keeper
  done
done
cat $O/keeper_ab.txt
