#!/bin/bash
# Round 5: the bench lines of every workload on one box (each step under its own limit; a step that hits it ends the set).
#   usage (on the box): bash tools/run_bench_set_r5.sh A|B [out dir]      A: ns, c2, c3, shards, white;  B: c5, c5s, c1, two-rank rehearsals
O=${2:-gpurun_out/r5q}; mkdir -p $O
b() { echo "$1 python bench.py $3 > $O/$2_bench.json 2> $O/$2_bench.err; tail -c 300 $O/$2_bench.err | tail -2"; }
if [ "$1" = A ]; then
  tools/gpu_seq.sh "$(b 420 ns "--loop em")" "$(b 300 c2 "--workload c2")" "$(b 420 c3 "--workload c3")" \
    "$(b 120 ns_rows125000 "--rows 125000 --steps 40")" "$(b 120 c3_rows125000 "--workload c3 --rows 125000 --steps 40")" \
    "$(b 120 ns_white "--model white")"
else
  tools/gpu_seq.sh "$(b 300 c5 "--workload c5 --steps 6 --warmup 1")" "$(b 300 c5s "--workload c5s --steps 6 --warmup 1")" "$(b 100 c1 "--workload c1 --steps 200")" \
    "$(b 120 ns_rows250000 "--rows 250000 --no-state-match")" "$(b 120 c3_rows250000 "--workload c3 --rows 250000 --no-state-match")" \
    "200 TSVGP_BENCH_BACKEND=gloo python bench.py --gpus 2 --rows 250000 --no-cpu-baseline > $O/selflaunch_2ranks_gloo_ns_rows250000_bench.json 2> $O/sl_ns.err; tail -2 $O/sl_ns.err" \
    "200 TSVGP_BENCH_BACKEND=gloo python bench.py --workload c3 --gpus 4 --rows 500000 --no-cpu-baseline > $O/selflaunch_4ranks_gloo_c3_rows500000_bench.json 2> $O/sl_c3.err; tail -2 $O/sl_c3.err"
fi
