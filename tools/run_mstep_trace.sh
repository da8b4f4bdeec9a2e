#!/bin/bash
# Round 5: rocprofv3 kernel statistics of tools/bench_mstep.py (the M-step gradient evaluation at N = 1e6, M = 1024).
# usage (on the box): bash tools/run_mstep_trace.sh [out dir]
R=$PWD; O=${1:-gpurun_out/r5w}; mkdir -p $O
python tools/bench_mstep.py > $O/mstep_plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_mstep -- python3 $R/tools/bench_mstep.py > $R/$O/kt_mstep.log 2>&1
f=$(find $R/$O/kt_mstep -name "*kernel_stats.csv" | head -1)
cp $f $R/$O/mstep_kernel_stats.csv
find $R/$O/kt_mstep -name "*.csv" -delete
cd $R
cat $O/mstep_plain.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/mstep_kernel_stats.csv")))
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms  x{r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
