#!/usr/bin/env python3
"""MFMA-operand write-after-read lint over the compiler's own assembly of the hand-laid kernels.

The hazard (tools/hazard_probe.hip measures it, profiles/r04_hazard_probe.txt records it): the A / B operand registers of a
``v_mfma_*`` are not all read when the instruction issues; an LDS (or memory) load whose data returns into one of them shortly
afterwards changes the product.  The hardware interlocks VALU writes, the LLVM hazard recognizer knows nothing of this one, and
with the instruction stream pinned by ``sched_barrier`` (panel1_kernel, syrk1_kernel, syrk1f_kernel) the register allocator is
free to hand a just-read operand register to the next fragment read.  This script finds every

    v_mfma  D, srcA, srcB, C   ...  <= WINDOW instructions later ...   ds_read* / global_load* / buffer_load*  -> overlaps srcA|srcB

following fall-through and branch edges backwards (loop back-edges included), and prints / returns the sites.

    python tools/isa_hazards.py [--window N] [--asm k.s]        exits 1 when a site is found inside the window
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "t-svgp_amd", "csrc", "tsvgp_kernels.hip")
KERNEL_TAGS = ("panel1_kernel", "syrk1_kernel", "syrk1f_kernel")  # the kernels whose streams are pinned by hand
ASYNC_WRITERS = ("ds_read", "ds_load", "global_load", "buffer_load", "flat_load", "scratch_load")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def compile_asm(path=None):
    """hipcc -S of the product source (cross-compiles on the CPU); returns the assembly text."""
    if path and os.path.exists(path):
        return open(path).read()
    with tempfile.TemporaryDirectory() as tmp:
        out = path or os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-I",
                        os.path.join(ROOT, "include"), SRC, "-o", out], check=True, capture_output=True)
        return open(out).read()


def regs(operand):
    """Register set of one operand text: 'v[8:9]' -> {('v', 8), ('v', 9)}; 'a5' -> {('a', 5)}; anything else -> empty."""
    out = set()
    for m in REG.finditer(operand):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def split_operands(text):
    """'v_mfma.. a[0:7], v[8:9], v[2:3], a[0:7] cbsz:1' -> ['a[0:7]', 'v[8:9]', 'v[2:3]', 'a[0:7] cbsz:1']."""
    parts = text.split(None, 1)
    return [] if len(parts) < 2 else [p.strip() for p in parts[1].split(",")]


def kernels(text):
    """{symbol: [(label, [instruction text, ...]), ...]} for the hand-laid kernels."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        name = m.group(1)
        if not any(t in name for t in KERNEL_TAGS):
            continue
        body = text[m.end():text.find("s_endpgm", m.end())].split("\n")
        blocks, cur = [], ("entry", [])
        for ln in body:
            t = ln.split(";")[0].strip()
            lab = re.match(r"^(\.LBB\d+_\d+):", ln.strip())
            if lab:
                blocks.append(cur)
                cur = (lab.group(1), [])
            elif t and not t.startswith("."):
                cur[1].append(t)
        blocks.append(cur)
        out[name] = blocks
    return out


def predecessors(blocks):
    """{block index: [predecessor block indices]}: fall-through (unless the previous block ends in s_branch) + branch targets."""
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    preds = {i: [] for i in range(len(blocks))}
    for i, (_, ins) in enumerate(blocks):
        falls = True
        for t in ins:
            if t.startswith(("s_cbranch", "s_branch")):
                tgt = t.split()[-1]
                if tgt in index:
                    preds[index[tgt]].append(i)
        if ins and ins[-1].startswith("s_branch"):
            falls = False
        if falls and i + 1 < len(blocks):
            preds[i + 1].append(i)
    return preds


def scan(text, window):
    """All (kernel, label, distance, mfmas_between, mfma text, writer text) with an asynchronous VGPR writer landing in srcA / srcB
    of an MFMA at most `window` instructions before it."""
    sites = []
    for name, blocks in kernels(text).items():
        preds = predecessors(blocks)

        def back(bi, pos, left, between, dest, writer, label, seen):
            """Walk backwards from instruction `pos` (exclusive) of block bi with `left` instructions of window to go."""
            _, ins = blocks[bi]
            j = pos - 1
            while j >= 0 and left > 0:
                t = ins[j]
                if t.startswith("v_mfma"):
                    ops = split_operands(t)
                    if len(ops) >= 3 and (regs(ops[1]) | regs(ops[2])) & dest:
                        sites.append((name, label, window - left + 1, between, t, writer))
                    between += 1
                j -= 1
                left -= 1
            if left > 0:
                for p in preds[bi]:
                    key = (p, left)
                    if key in seen:
                        continue
                    seen.add(key)
                    back(p, len(blocks[p][1]), left, between, dest, writer, label, seen)

        for bi, (label, ins) in enumerate(blocks):
            for pos, t in enumerate(ins):
                if t.startswith(ASYNC_WRITERS) and "lds" not in t.split()[0]:
                    ops = split_operands(t)
                    dest = regs(ops[0]) if ops else set()
                    dest = {r for r in dest if r[0] == "v"}
                    if dest:
                        back(bi, pos, window, 0, dest, t, label, set())
    return sites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=8, help="instructions between the MFMA and the load that count as a site")
    ap.add_argument("--fail-within", type=int, default=None, help="exit 1 when a site is at most this many instructions apart")
    ap.add_argument("--asm", default=None, help="reuse / keep the assembly at this path")
    args = ap.parse_args()
    text = compile_asm(args.asm)
    sites = scan(text, args.window)
    ks = kernels(text)
    print(f"{len(ks)} hand-laid kernel instantiations, "
          f"{sum(sum(t.startswith('v_mfma') for _, ins in b for t in ins) for b in ks.values())} MFMAs, window {args.window} instructions")
    by = {}
    for s in sites:
        by.setdefault(s[0], []).append(s)
    for name in sorted(ks):
        found = by.get(name, [])
        hist = {}
        for s in found:
            hist[s[2]] = hist.get(s[2], 0) + 1
        print(f"  {name}: {len(found)} site(s)" + (f", by distance {dict(sorted(hist.items()))}" if found else ""))
        for s in sorted(found, key=lambda s: s[2])[:6]:
            print(f"      {s[1]}: distance {s[2]} ({s[3]} MFMAs between)   {s[4]}   ->   {s[5]}")
    lim = args.fail_within
    if lim is not None and any(s[2] <= lim for s in sites):
        sys.exit(1)


if __name__ == "__main__":
    main()
