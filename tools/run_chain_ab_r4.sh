#!/bin/bash
# Round 4: the replicated M x M chain, old paths against new ones on ONE box (env toggles select the round-3 paths of this build):
#   old = TSVGP_POTRF_SOLVE=0 (inverse factors + GEMM for D) TSVGP_SITE_UPDATE=0 (site_target + gemv + elementwise launches)
#   new = the defaults (factor-and-solve for W and K_uu + jitter I in one batch, fused site update); K(Z, Z) in front of the N-sized fill in both
# bench lines alternate over (old, new) for each (workload, rows); then the rocprofv3 M x M timelines of the new paths.
# usage (on the box): bash tools/run_chain_ab_r4.sh [out dir]
R=$PWD; O=${1:-gpurun_out/r4g}; mkdir -p $O; : > $O/chain_ab.txt
line() {  # workload rows steps tag
  python bench.py --workload $1 --rows $2 --steps $3 --no-elbo-match --no-cpu-baseline --no-side-lines --no-state-match 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$4', '$1', $2, 'ms/step', d['ms_per_step'], 'hipgraph' if d.get('hipgraph') and d['hipgraph'].get('headline_mode') == 'hipGraph replay' else 'eager',
      *[f'{a} {k[b][\"avg_ms\"]}' for a, b in (('potrf', 'tsvgp_potrf'), ('moments', 'tsvgp_moments'), ('site_accum', 'tsvgp_site_accum'), ('fill', 'tsvgp_se_fill'), ('kuu', 'tsvgp_se_fill(Kuu)')) if b in k])" >> $O/chain_ab.txt
}
for rep in 1 2; do
  for spec in "ns 125000 40" "c3 125000 40" "ns 1000000 20"; do
    set -- $spec
    TSVGP_POTRF_SOLVE=0 TSVGP_SITE_UPDATE=0 TSVGP_SITE_BETA=0 line $1 $2 $3 old
    line $1 $2 $3 new
  done
done
cat $O/chain_ab.txt
cd /tmp && export TMPDIR=/tmp
for spec in "ns 125000" "c3 125000" "ns 1000000"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$1_$2 -- python3 $R/bench.py --workload $1 --rows $2 --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines --no-state-match > $R/$O/kt_$1_$2.log 2>&1
  f=$(find $R/$O/kt_$1_$2 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/mxm_timeline.py $f 4 --list > $R/$O/mxm_timeline_$1_rows$2.txt 2>&1 || true
  find $R/$O/kt_$1_$2 -name "*kernel_trace.csv" -delete
done
cd $R
echo done
