#!/bin/bash
# Round 5: rows per workgroup of the N-sized K(X, Z) fill (TSVGP_FILL_ROWS_BLK) -- shorter-lived fill workgroups free CU slots sooner
# for the factorisation kernels that run beside them.  Bench lines on ONE box.  usage: bash tools/run_fill_rows_ab.sh [out] [workload]
R=$PWD; O=${1:-gpurun_out/r5j}; W=${2:-ns}; mkdir -p $O; : > $O/fill_rows_$W.txt
for rep in 1 2; do
  for rb in 64 32 16 8; do
    TSVGP_FILL_ROWS_BLK=$rb python bench.py --workload $W --steps 20 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null > $O/line_$W_$rb.json
    python -c "
import json
d = json.loads([l for l in open('$O/line_$W_$rb.json').read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('rows_blk $rb', '$W', 'ms/step', d['ms_per_step'], *[f'{a} {k[b][\"avg_ms\"]} (max {k[b][\"max_ms\"]})' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill')) if b in k])" >> $O/fill_rows_$W.txt
  done
done
cat $O/fill_rows_$W.txt
