"""Bitwise repeatability of the hand-laid kernels at a size where every CU runs several workgroups (tools/determinism_check.py):
moments / trmm in the upper and lower form and the site sums, fp64 and fp32, four launches each on one input -- all outputs
bit-identical, the first against a torch fp64 reference.  A timing-dependent fault (the lab kernel of round 3 had lanes corrupted
differently from run to run: inline-asm memory operations that nothing pads or counts -- the rules of tests/test_isa_lint.py)
fails the first half of that."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_repeated_launches_are_bit_identical_and_right():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "determinism_check.py")
    spec = importlib.util.spec_from_file_location("determinism_check", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    assert mod.check(rows=66000, M_=1024, reps=4, out=lines.append) == 0, "\n".join(lines)
    assert len(lines) == 6
