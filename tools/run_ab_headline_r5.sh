#!/bin/bash
# Round 5: the headline workload (ns, N = 1e6) with round 4's block step of the factorisation (TSVGP_POTRF_DIAG_V1=1) against the
# default, alternating on ONE box.  usage: bash tools/run_ab_headline_r5.sh [out dir] [workload] [rows]
R=$PWD; O=${1:-gpurun_out/r5g}; W=${2:-ns}; ROWS=${3:-1000000}; mkdir -p $O; : > $O/ab_$W.txt
line() {
  python bench.py --workload $W --rows $ROWS --steps 20 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null > $O/line_$1_$W.json
  python -c "
import json
d = json.loads([l for l in open('$O/line_$1_$W.json').read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$1', '$W', 'ms/step', d['ms_per_step'], *[f'{a} {k[b][\"avg_ms\"]} (max {k[b][\"max_ms\"]})' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill')) if b in k])" >> $O/ab_$W.txt
}
for rep in 1 2; do
  TSVGP_POTRF_DIAG_V1=1 line r4step
  line new
done
cat $O/ab_$W.txt
