#!/bin/bash
# Round 5: the fill's stores as write-through streaming stores (TSVGP_FILL_STORE=wt: global_store ... sc0 sc1 nt) against the default
# (nt), alternating on ONE box: the fill alone, then bench lines.   usage (on the box): bash tools/run_fill_store_ab.sh [out dir]
O=${1:-gpurun_out/r6d}; mkdir -p $O; : > $O/fill_store_ab.txt
line() {  # workload rows steps tag
  python bench.py --workload $1 --rows $2 --steps $3 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null > $O/line_$4_$1_$2.json
  python -c "
import json
d = json.loads([l for l in open('$O/line_$4_$1_$2.json').read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$4', '$1', $2, 'ms/step', d['ms_per_step'], *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill')) if b in k])" >> $O/fill_store_ab.txt
}
for rep in 1 2; do
  for spec in "ns 1000000 20" "ns 125000 40" "c3 1000000 20" "c2 1000000 20"; do
    set -- $spec
    line $1 $2 $3 nt
    TSVGP_FILL_STORE=wt line $1 $2 $3 wt
  done
done
cat $O/fill_store_ab.txt
