#!/bin/bash
# Round-3 evidence on the GPU box (one gpurun call): rocprofv3 kernel trace + stats of the headline workload and of one rank's
# share of an 8-way shard (M x M timelines), then the PMC passes (tools/pmc_passes.sh) and profiles/hbm_traffic.json from them.
# usage: bash tools/run_evidence_r3.sh [trace|pmc|all]     output under gpurun_out/r3i (copied into profiles/ afterwards)
set -e
R=$PWD; O=gpurun_out/r3i; mkdir -p $O
what=${1:-all}
if [ "$what" = trace ] || [ "$what" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines > $R/$O/kt.log 2>&1
  cd $R
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  python tools/mxm_timeline.py $f 4 --list > $O/mxm_timeline.txt 2>&1 || true
  python tools/trace_vs_events.py $f $O/kt.log > $O/trace_vs_events.txt 2>&1 || true
  find $O/kt -name "*kernel_trace.csv" -delete
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt125 -- python3 $R/bench.py --rows 125000 --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines > $R/$O/kt125.log 2>&1
  cd $R
  f=$(find $O/kt125 -name "*kernel_trace.csv" | head -1)
  python tools/mxm_timeline.py $f 4 --list > $O/mxm_timeline_125k.txt 2>&1 || true
  find $O/kt125 -name "*kernel_trace.csv" -delete
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  bash tools/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
  python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1 || true
  python tools/make_hbm_traffic.py $O/pmc ns --source "profiles/r03_v2_ns_pmc_summary.txt" > $O/traffic.txt 2>&1 || true
  cp profiles/hbm_traffic.json $O/hbm_traffic.json
  find $O/pmc -name "*.csv" -size +5M -delete
fi
echo done
