#!/bin/bash
# Round 4: priority class of the side stream that runs the K(X, Z) fill beside the M x M prelude (TSVGP_FILL_PRIORITY = normal | low | high),
# bench lines alternating on ONE box for the 125 000-row shards (fill hidden behind the prelude), N = 1e6 (fill exposed) and C2.
# usage (on the box): bash tools/run_fill_priority_ab.sh [out dir]
O=${1:-gpurun_out/r4w}; mkdir -p $O; : > $O/fill_priority_ab.txt
line() {  # workload rows steps tag
  timeout -k 10 200 python bench.py --workload $1 --rows $2 --steps $3 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>$O/err.txt | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$4', '$1', $2, 'ms/step', d['ms_per_step'], 'hipgraph' if d.get('hipgraph') and d['hipgraph'].get('headline_mode') == 'hipGraph replay' else 'eager',
      *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill'), ('kuu', 'tsvgp_se_fill(Kuu)')) if b in k])" >> $O/fill_priority_ab.txt || { tail -5 $O/err.txt; exit 1; }
}
for rep in 1 2; do
  for spec in "ns 125000 40" "c3 125000 40" "ns 1000000 20" "c2 1000000 30"; do
    set -- $spec
    for pr in normal low high; do
      TSVGP_FILL_PRIORITY=$pr line $1 $2 $3 $pr || exit 1
    done
  done
done
cat $O/fill_priority_ab.txt
