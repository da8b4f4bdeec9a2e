#!/usr/bin/env python3
"""Lint of the compiler's own assembly of the hand-laid kernels: the properties the hand-written instruction streams rely on
and that nothing else checks (the LLVM hazard recognizer does not look inside ``asm volatile``; a compiler update can move
registers and wait states silently).  Used by tests/test_isa_lint.py (CPU: hipcc cross-compiles) and from the command line.

Rules (``check_rules``):
  m0          an LDS-DMA (``global_load_lds_*``) reads M0: the instruction in front of it must not be the one that writes M0
              (one wait state; the kernels put ``s_nop 0`` inside the asm statement).
  sgpr-vmem   a scalar register written by a VECTOR instruction (``v_readfirstlane_b32`` of the DMA base addresses) needs five
              wait states in front of a memory instruction that reads it as its scalar base; the kernels make the bases scalar
              at the START of a chunk, three k-steps before use.
  scratch     no scratch access in any basic block that holds MFMAs of the one-workgroup-per-CU kernels (512 registers per
              wave: the whole point of that launch bound; a scratch access in the stream also waits for the LDS-DMA in flight).

And one REPORT (``scan``), no longer a rule: MFMA-operand write-after-read sites.  Round 3 believed that a ``ds_read`` landing
in an A / B operand register of a ``v_mfma_f64_16x16x4_f64`` issued just before it corrupted results, and kept such reads apart
by convention.  Round 4 measured it (tools/hazard_probe.hip, profiles/r04_hazard_probe.txt): no window exists -- not with an
idle pipe, not behind one to three MFMAs in flight, not in a dependent accumulator chain, AGPR or VGPR accumulators, fp64 or
fp32 -- and a build of panel1_kernel with the A fragments read FIRST and the registers released at once (-DTSVGP_HAZARD_AFIRST:
100-190 sites at distance 1 in every instantiation) passes every kernel parity test and the bitwise-repeatability check.  The
operands are read when the MFMA issues.  The scan stays as a report so that a future anomaly can be correlated with it.

    python tools/isa_hazards.py [--window N] [--asm k.s]        exits 1 when a RULE is violated
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


SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def wait_states(t):
    """Wait states an instruction provides to what follows it: s_nop N gives N + 1, everything else 1."""
    m = re.match(r"s_nop\s+(\d+)", t)
    return int(m.group(1)) + 1 if m else 1


def all_kernels(text):
    """Every kernel of the file that issues LDS-DMA or belongs to the hand-laid set."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        end = text.find("s_endpgm", m.end())
        body = text[m.end():end]
        if "global_load_lds" in body or any(t in m.group(1) for t in KERNEL_TAGS):
            out[m.group(1)] = body
    return out


def check_rules(text):
    """[(rule, kernel, label, message)] for every violated rule."""
    bad = []
    ks = kernels(text)
    extra = {k: None for k in all_kernels(text) if k not in ks}
    for name in list(ks) + list(extra):
        if name in ks:
            blocks = ks[name]
        else:  # a kernel outside the hand-laid set that issues LDS-DMA: same parsing
            blocks = _blocks_of(text, name)
        preds = predecessors(blocks)

        def walk_back(bi, pos, budget, visit, seen):
            """visit(instruction) for the instructions in front of (bi, pos), nearest first, while `budget` wait states last;
            visit returns True to stop that path."""
            _, ins = blocks[bi]
            j = pos - 1
            while j >= 0 and budget > 0:
                if visit(ins[j], budget):
                    return
                budget -= wait_states(ins[j])
                j -= 1
            if budget > 0:
                for p_ in preds[bi]:
                    if (p_, budget) not in seen:
                        seen.add((p_, budget))
                        walk_back(p_, len(blocks[p_][1]), budget, visit, seen)

        one_wg = ("panel1_kernel" in name or "syrk1_kernel" in name)
        for bi, (label, ins) in enumerate(blocks):
            if one_wg and any(t.startswith("v_mfma") for t in ins) and any(t.startswith("scratch_") for t in ins):
                bad.append(("scratch", name, label, "scratch access in a basic block that holds MFMAs"))
            for pos, t in enumerate(ins):
                op = t.split()[0]
                if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in t):
                    def m0_writer(prev, budget, t=t, label=label):
                        if re.match(r"s_\w+\s+m0\b", prev):
                            bad.append(("m0", name, label, f"'{prev}' directly in front of '{t}'"))
                        return True  # only the instruction directly in front matters (one wait state)
                    walk_back(bi, pos, 1, m0_writer, set())
                if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "global_atomic")):
                    ops_ = split_operands(t)
                    used = set()
                    for o in ops_:
                        if re.match(r"^s(\[|\d)", o.strip()):
                            used |= sregs(o)
                    if used:
                        def valu_writer(prev, budget, t=t, label=label, used=used):
                            if prev.startswith(("v_readfirstlane", "v_readlane")):
                                dst = sregs(split_operands(prev)[0])
                                if dst & used:
                                    bad.append(("sgpr-vmem", name, label,
                                                f"'{prev}' only {5 - budget + 1} wait state(s) in front of '{t}' (needs 5)"))
                                    return True
                            return False
                        walk_back(bi, pos, 5, valu_writer, set())
    return bad


def _blocks_of(text, name):
    m = re.search(r"^" + re.escape(name) + r":", text, re.M)
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
    return blocks


def vgpr_counts(text):
    """{kernel symbol: VGPRs} from the metadata of the assembly (.vgpr_count)."""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\s*\n(?:.*\n)*?\s*\.vgpr_count:\s+(\d+)", text):
        out[m.group(1)] = int(m.group(2))
    return out


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
    bad = check_rules(text)
    print(f"rules (m0 wait state, VALU-written SGPR -> VMEM base, no scratch in the MFMA stream): {len(bad)} violation(s)")
    for b in bad[:40]:
        print("   ", *b)
    lim = args.fail_within
    if bad or (lim is not None and any(s[2] <= lim for s in sites)):
        sys.exit(1)


if __name__ == "__main__":
    main()
