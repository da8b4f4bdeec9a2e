#!/bin/bash
# Collects the per-round evidence on the GPU box: bench lines of every workload, the rocprofv3 kernel trace and the PMC
# passes of the headline workload, under gpurun_out/$EVIDENCE_DIR (copied into profiles/ by hand afterwards).
# usage: EVIDENCE_DIR=r02 bash tools/run_evidence.sh <stage>     stage 1: ns, c2, c3, c1 bench lines; 2: c5; 2b: c5s, white,
#        125 000-row shard; 3: kernel trace + M x M timeline + PMC passes   (one gpurun call each: 20 minutes at most)
set -e
R=$PWD
O=gpurun_out/${EVIDENCE_DIR:-r02}
mkdir -p $O
case "$1" in
1)
  python bench.py > $O/ns_bench.json 2> $O/ns_bench.err; echo ns done
  python bench.py --workload c2 > $O/c2_bench.json 2> $O/c2.err; echo c2 done
  python bench.py --workload c3 > $O/c3_bench.json 2> $O/c3.err; echo c3 done
  python bench.py --workload c1 > $O/c1_bench.json 2> $O/c1.err; echo c1 done
  ;;
2)
  python bench.py --workload c5 --steps 5 --warmup 2 --elbo-budget 60 > $O/c5_bench.json 2> $O/c5.err; echo c5 done
  ;;
2b)
  python bench.py --workload c5s --steps 5 --warmup 2 --elbo-budget 60 > $O/c5s_bench.json 2> $O/c5s.err; echo c5s done
  python bench.py --model white --steps 10 --warmup 3 --no-cpu-baseline > $O/ns_white_bench.json 2> $O/white.err; echo white done
  python bench.py --rows 125000 --steps 40 --warmup 5 --no-cpu-baseline --no-elbo-match --no-side-lines > $O/ns_rows125000_bench.json 2> $O/rows.err
  ;;
3)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines > $R/$O/kt.log 2>&1
  cd $R
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  python tools/mxm_timeline.py $f 4 --list > $O/mxm_timeline.txt 2>&1 || true
  python tools/trace_vs_events.py $f $O/kt.log > $O/trace_vs_events.txt 2>&1 || true
  find $O/kt -name "*kernel_trace.csv" -delete   # tens of MB; the stats file and the timeline are what is kept
  echo kt done
  bash tools/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
  python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1 || true
  find $O/pmc -name "*.csv" -size +5M -delete
  ;;
esac
echo stage $1 done
