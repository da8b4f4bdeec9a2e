#!/usr/bin/env python3
"""profiles/elbo_1gpu.json from ONE-GPU bench lines (bench.py --gpus 1) measured on MI355X: the reference value of
`elbo_vs_1gpu` in multi-rank bench lines.

    python tools/make_elbo_1gpu.py profiles/r03_*_bench.json [...]

Each input is one JSON line of bench.py.  Keyed by model ("tsvgp" / "white") and workload ("ns", "c2", ...; "ns@125000"
for a --rows run).  Lines with n_gpus != 1 are refused; a later file overrides an earlier one with the same key."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"gaussian_regression_N1e6_M1024_D8_P1": "ns", "gaussian_regression_N1e6_M512_D8_P1": "c2",
         "bernoulli_probit_N1e6_M1024_D16_P1": "c3", "gaussian_multioutput_N1e6_M1024_D8_P8_separate_kernels": "c5",
         "gaussian_multioutput_N1e6_M1024_D8_P8_shared_kernel": "c5s", "gaussian_1d_N1000_M32": "c1"}
FULL_N = {"ns": 1_000_000, "c2": 1_000_000, "c3": 1_000_000, "c5": 1_000_000, "c5s": 1_000_000, "c1": 1000}


def main(paths):
    out = {"tsvgp": {}, "white": {}}
    for path in paths:
        line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        if d.get("n_gpus") != 1:
            sys.exit(f"{path}: n_gpus = {d.get('n_gpus')}, not a one-GPU line")
        name = d["config"]["workload"]
        kind = "white" if "[t_SVGP_white]" in name else "tsvgp"
        wl = NAMES[re.split(r" [\(\[]", name)[0]]
        key = wl if d["config"]["N"] == FULL_N[wl] else f"{wl}@{d['config']['N']}"
        taken = d.get("steps_before_elbo", d["warmup"] + d["steps"])
        out[kind][key] = {"elbo_after_steps": d["elbo_after_steps"], "steps_taken": taken,
                          "source": os.path.relpath(os.path.abspath(path), ROOT)}
    dst = os.path.join(ROOT, "profiles", "elbo_1gpu.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(f"wrote {dst}: {sum(len(v) for v in out.values())} entries")


if __name__ == "__main__":
    main(sys.argv[1:])
