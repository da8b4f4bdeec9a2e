#!/bin/bash
# Round 4: one long run per workload on the last build with the parity half (state match / ELBO match against the oracle) taken on the
# FINAL state: ns 200 steps, c3 200, c2 300, a 125 000-row ns shard 400 replayed steps.   usage (on the box): bash tools/run_soak_r4.sh [out dir]
O=${1:-gpurun_out/r4soak}; mkdir -p $O
b() { echo "$1 python bench.py $3 --warmup 5 --no-side-lines --no-cpu-baseline > $O/$2.json 2> $O/$2.err; tail -1 $O/$2.err"; }
tools/gpu_seq.sh "$(b 300 ns "--steps 200")" "$(b 300 c3 "--workload c3 --steps 200")" "$(b 200 c2 "--workload c2 --steps 300")" "$(b 120 ns_rows125000 "--rows 125000 --steps 400")"
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for n in ("ns", "c3", "c2", "ns_rows125000"):
    try:
        d = json.loads([l for l in open(f"{o}/{n}.json").read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(n, "no line:", e); continue
    sm = d.get("state_match") or {}
    print(n, "steps", d["steps"], "value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "moments", d["roofline"]["frac"],
          "| elbo rel", d["elbo_match"]["rel"], "| state after one more step:", sm.get("state_after_step_max_rel_err"))
PY
