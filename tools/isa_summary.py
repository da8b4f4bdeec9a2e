#!/usr/bin/env python3
"""Per-basic-block instruction counts of the hand-laid kernels from the compiler's own assembly (hipcc -S): where the MFMAs
are, and that no scratch access, no v_accvgpr copy sits in the chunk stream.  Also prints one steady-state block in full.
    python tools/isa_summary.py > profiles/r03_isa_summary.txt        (runs on the CPU: hipcc cross-compiles)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "t-svgp_amd", "csrc", "tsvgp_kernels.hip")
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-I",
                    os.path.join(ROOT, "include"), src, "-o", asm], check=True)
    text = open(asm).read()
KERNELS = [("panel1_kernelIdLi1", "panel1_kernel<double, MOMENTS>  (tsvgp_moments_f64, upper form)"),
           ("panel1_kernelIfLi1", "panel1_kernel<float, MOMENTS>   (tsvgp_moments_f32)"),
           ("panel1_kernelIdLi0", "panel1_kernel<double, STORE>    (tsvgp_trmm_f64, upper form)"),
           ("syrk1_kernel", "syrk1_kernel  (tsvgp_site_accum_f64)"), ("syrk1f_kernel", "syrk1f_kernel (tsvgp_site_accum_f32)")]
COLS = ("v_mfma", "scratch_", "v_accvgpr", "ds_read", "global_load_lds", "s_waitcnt", "s_barrier", "other")
print("columns per basic block:", ", ".join(COLS), "  (blocks without MFMA, scratch or accvgpr instructions are left out)\n")
first_dump = True
for tag, title in KERNELS:
    m = re.search(r"^(_Z\w*" + tag + r"\w*):", text, re.M)
    if not m:
        print(title, ": not found"); continue
    body = text[m.end():text.find("s_endpgm", m.end())].split("\n")
    res = re.search(r"\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text[text.find(".name:           " + m.group(1)):][:3000], re.S)
    blocks, cur = [], ["entry", [], ""]
    for ln in body:
        t = ln.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", t)
        if lab:
            blocks.append(cur); cur = [lab.group(1), [], (lab.group(2) or "")]
        elif t and not t.startswith((";", ".")):
            cur[1].append(t)
    blocks.append(cur)
    print(f"== {title}: {sum(len(b[1]) for b in blocks)} instructions, {len(blocks)} blocks")
    tot = dict.fromkeys(COLS, 0)
    steady = None
    for name, ins, note in blocks:
        c = dict.fromkeys(COLS, 0)
        for i in ins:
            for k in COLS[:-1]:
                if i.startswith(k) or (k == "v_accvgpr" and "v_accvgpr" in i):
                    c[k] += 1; break
            else:
                c["other"] += 1
        for k in COLS: tot[k] += c[k]
        if c["v_mfma"] or c["scratch_"] or c["v_accvgpr"]:
            depth = re.search(r"Depth=(\d)", note)
            print(f"   {name:12s} depth {depth.group(1) if depth else '0'}  " + "  ".join(f"{c[k]:4d}" for k in COLS))
            if c["v_mfma"] >= 128 and "Inner Loop" in note and steady is None:
                steady = (name, ins)
    print("   total" + " " * 16 + "  ".join(f"{tot[k]:4d}" for k in COLS))
    in_stream = sum(1 for name, ins, note in blocks if any(i.startswith("v_mfma") for i in ins) and any(i.startswith("scratch_") for i in ins))
    print(f"   blocks holding both MFMAs and scratch accesses: {in_stream}\n")
    if steady and first_dump:
        first_dump = False
        print(f"   --- steady-state block {steady[0]} of this kernel in full ({len(steady[1])} instructions: two full k-chunks) ---")
        for i in steady[1]:
            print("      " + i.split(";")[0].rstrip())
        print()
