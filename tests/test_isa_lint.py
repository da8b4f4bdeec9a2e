"""The generated code of the hand-laid kernels, checked (CPU: hipcc cross-compiles gfx950).

panel1_kernel / syrk1_kernel / syrk1f_kernel pin their instruction order with sched_barrier and move their operands with
inline-asm LDS-DMA; the LLVM hazard recognizer does not look inside ``asm volatile`` and a compiler update can move registers
and wait states silently.  tools/isa_hazards.py parses the compiler's assembly; this file holds it to the rules the streams
rely on (see that file's header and DESIGN.md section 4), and checks the lint itself on hand-made violations.

The MFMA-operand write-after-read "hazard" of round 3 is NOT a rule: tools/hazard_probe.hip and a build with the suspect read
order (profiles/r04_hazard_probe.txt) showed that no such window exists on gfx950; the scan is kept as a report only.
"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lint():
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "tools", "isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake(body, name="_ZN12_GLOBAL__N_113panel1_kernelIdLi1ELi1ELi1EEEvNS_9PanelArgsIT_EE"):
    return f"\t.text\n{name}:\n" + "\n".join("\t" + ln for ln in body) + "\n\ts_endpgm\n"


def test_rules_hold_on_the_product(lint, product_asm):
    ks = lint.kernels(product_asm)
    # every instantiation the C-ABI launches is there: 8 x panel1_kernel (type x mode x triangle), syrk1_kernel, syrk1f_kernel
    assert sum("panel1_kernel" in k for k in ks) == 8 and any("syrk1_kernel" in k for k in ks) and any("syrk1f_kernel" in k for k in ks)
    dma = sum(t.startswith("global_load_lds") for b in ks.values() for _, ins in b for t in ins)
    mfma = sum(t.startswith("v_mfma") for b in ks.values() for _, ins in b for t in ins)
    assert dma > 500 and mfma > 5000  # the parser sees the streams (round 4: 1354 LDS-DMA issues, 11008 MFMAs)
    bad = lint.check_rules(product_asm)
    assert not bad, "\n".join(" ".join(map(str, b)) for b in bad[:20])


def test_spills_stay_outside_the_chunk_stream(lint, product_asm):
    """__launch_bounds__(256, 1) gives a wave 512 registers; the chunk streams were laid out so that nothing in them touches
    scratch (a scratch access in the stream also waits for the LDS-DMA in flight).  The fp64 upper-form moments kernel does keep
    ~80 panel-invariant values in scratch -- stored once in its prologue, reloaded in the per-panel epilogue (the likelihood
    map) -- which is fine; what must not happen is a scratch access inside a loop that holds MFMAs (rule "scratch" above covers
    blocks; this covers the count: a jump in it means the allocator has started spilling the stream's own values)."""
    for name, blocks in lint.kernels(product_asm).items():
        n = sum(t.startswith("scratch_") for _, ins in blocks for t in ins)
        assert n <= 120, f"{name}: {n} scratch instructions (round 4: at most 84, all in prologue / epilogue blocks)"


def test_lint_flags_a_missing_m0_wait_state(lint):
    ok = _fake(["s_mov_b32 m0, s5", "s_nop 0", "global_load_lds_dwordx4 v1, s[8:9]"])
    bad = _fake(["s_mov_b32 m0, s5", "global_load_lds_dwordx4 v1, s[8:9]"])
    assert not lint.check_rules(ok)
    found = lint.check_rules(bad)
    assert len(found) == 1 and found[0][0] == "m0"


def test_lint_flags_a_scalar_base_fresh_from_a_vector_instruction(lint):
    near = _fake(["v_readfirstlane_b32 s8, v3", "s_mov_b32 m0, s5", "s_nop 0", "global_load_lds_dwordx4 v1, s[8:9]"])
    far = _fake(["v_readfirstlane_b32 s8, v3", "s_nop 4", "s_mov_b32 m0, s5", "s_nop 0", "global_load_lds_dwordx4 v1, s[8:9]"])
    other = _fake(["v_readfirstlane_b32 s20, v3", "s_mov_b32 m0, s5", "s_nop 0", "global_load_lds_dwordx4 v1, s[8:9]"])
    across = ("\t.text\n_ZN12_GLOBAL__N_112syrk1_kernelENS_8SyrkArgsIdEE:\n\tv_readfirstlane_b32 s9, v3\n\ts_branch .LBB0_2\n"
              ".LBB0_1:\n\ts_nop 7\n.LBB0_2:\n\ts_nop 0\n\tglobal_load_dwordx2 v[4:5], v1, s[8:9]\n\ts_endpgm\n")
    assert [b[0] for b in lint.check_rules(near)] == ["sgpr-vmem"]
    assert not lint.check_rules(far) and not lint.check_rules(other)
    assert [b[0] for b in lint.check_rules(across)] == ["sgpr-vmem"]  # followed over the branch edge, not the fall-through


def test_lint_flags_scratch_in_the_mfma_stream(lint):
    bad = _fake(["v_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]", "scratch_load_dwordx2 v[2:3], off, off offset:16"])
    ok = _fake(["scratch_load_dwordx2 v[2:3], off, off offset:16", "s_branch .LBB1_1", ".LBB1_1:",
                "v_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]"])
    assert [b[0] for b in lint.check_rules(bad)] == ["scratch"]
    assert not lint.check_rules(ok)


def test_operand_overwrite_scan_reports_sites(lint):
    """The report (not a rule, see the module docstring): a ds_read landing in srcB of an MFMA two instructions earlier."""
    txt = _fake(["v_mfma_f64_16x16x4_f64 a[0:7], v[8:9], v[2:3], a[0:7]", "ds_read_b64 v[38:39], v70", "ds_read_b64 v[2:3], v70 offset:8"])
    sites = lint.scan(txt, 4)
    assert len(sites) == 1 and sites[0][2] == 2 and sites[0][3] == 0
    assert not lint.scan(txt, 1)
