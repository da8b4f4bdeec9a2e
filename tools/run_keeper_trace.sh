#!/bin/bash
# Round 5: kernel traces of the replayed 125 000-row step with the clock keeper off / on.  usage: bash tools/run_keeper_trace.sh [out] [workload] [rows] [env assignments for the "on" leg ...]
R=$PWD; O=${1:-gpurun_out/r5u}; W=${2:-ns}; ROWS=${3:-125000}; shift 3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
leg() {  # tag
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$1 -- python3 $R/bench.py --workload $W --rows $ROWS --steps 30 --warmup 5 --no-cpu-baseline --no-elbo-match --no-side-lines --no-state-match > $R/$O/kt_$1.log 2>&1
  f=$(find $R/$O/kt_$1 -name "*kernel_trace.csv" | head -1)
  echo "== $1 ($W, $ROWS rows) ==" >> $R/$O/keeper_trace.txt
  grep -h '^{' $R/$O/kt_$1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('bench line under the tracer: ms/step', d['ms_per_step'])" >> $R/$O/keeper_trace.txt
  python3 $R/tools/step_trace.py $f 20 >> $R/$O/keeper_trace.txt 2>&1
  find $R/$O/kt_$1 -name "*.csv" -delete
}
: > $R/$O/keeper_trace.txt
leg off
TSVGP_CLOCK_KEEPER=-1 leg on
cat $R/$O/keeper_trace.txt
