#!/bin/bash
# Round 5: the replicated M x M chain at one rank's share of an 8-way shard (125 000 rows): bench lines for `ns` and `c3`
# (TSVGP_POTRF_DIAG_V1=1 = round 4's block step beside the default), then the rocprofv3 M x M timelines of an eager step.
# usage (on the box): bash tools/run_chain_r5.sh [out dir] [tag]
R=$PWD; O=${1:-gpurun_out/r5b}; T=${2:-v1}; mkdir -p $O; : > $O/chain_ab.txt
line() {  # workload rows steps tag
  python bench.py --workload $1 --rows $2 --steps $3 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null > $O/line_$4_$1_$2.json
  python -c "
import json, sys
d = json.loads([l for l in open('$O/line_$4_$1_$2.json').read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$4', '$1', $2, 'ms/step', d['ms_per_step'], 'hipgraph' if d.get('hipgraph') and d['hipgraph'].get('headline_mode') == 'hipGraph replay' else 'eager',
      *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill'), ('kuu', 'tsvgp_se_fill(Kuu)')) if b in k])" >> $O/chain_ab.txt
}
for rep in 1 2; do
  for spec in "ns 125000 40" "c3 125000 40"; do
    set -- $spec
    TSVGP_POTRF_DIAG_V1=1 line $1 $2 $3 r4step
    line $1 $2 $3 new
  done
done
cat $O/chain_ab.txt
cd /tmp && export TMPDIR=/tmp
for spec in "ns 125000" "c3 125000"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$1_$2 -- python3 $R/bench.py --workload $1 --rows $2 --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines --no-state-match > $R/$O/kt_$1_$2.log 2>&1
  f=$(find $R/$O/kt_$1_$2 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/mxm_timeline.py $f 4 --list > $R/$O/r05_${T}_$1_mxm_timeline_rows$2.txt 2>&1 || true
  find $R/$O/kt_$1_$2 -name "*kernel_trace.csv" -delete
done
cd $R
echo done
