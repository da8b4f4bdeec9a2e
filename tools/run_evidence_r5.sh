#!/bin/bash
# Round-5 evidence on the GPU box (one gpurun call): rocprofv3 kernel trace + stats of the headline workload with the trace-vs-events
# comparison of the SAME run, then the PMC passes (tools/pmc_passes.sh: separate rocprofv3 --pmc runs, no tracing beside them) and
# profiles/hbm_traffic.json regenerated from them.   usage: bash tools/run_evidence_r4.sh [trace|pmc|all] [out dir] [build tag: v1]
set -e
R=$PWD; O=${2:-gpurun_out/r5r}; V=${3:-v1}; mkdir -p $O
what=${1:-all}
if [ "$what" = trace ] || [ "$what" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines --no-state-match > $R/$O/kt.log 2>&1
  cd $R
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  python tools/trace_vs_events.py $f $O/kt.log > $O/trace_vs_events.txt 2>&1 || true
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  find $O/kt -name "*kernel_trace.csv" -delete
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  bash tools/pmc_passes.sh $O/pmc --no-state-match > $O/pmc.log 2>&1
  python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1 || true
  python tools/make_hbm_traffic.py $O/pmc ns --source "profiles/r05_${V}_ns_pmc_summary.txt" > $O/traffic.txt 2>&1 || true
  cp profiles/hbm_traffic.json $O/hbm_traffic.json
  find $O/pmc -name "*.csv" -size +5M -delete
fi
echo done
