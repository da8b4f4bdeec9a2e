"""(CPU) The lane-level NumPy model of the round-5 Cholesky kernels (tools/emul_diag2.py: potrf_diag2_kernel's tile dataflow
with its barrier placement, chol_panel2_kernel's substitution) against numpy.linalg -- the index algebra the HIP kernels of
t-svgp_amd/csrc/tsvgp_chol.hip are written from (they replace tf.linalg.cholesky / triangular_solve of reference
src/util.py:376-389, src/models/tsvgp.py:270, :300)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emul():
    spec = importlib.util.spec_from_file_location("emul_diag2", os.path.join(ROOT, "tools", "emul_diag2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tile_dataflow_and_substitution_panel_match_numpy():
    E = _emul()
    rng = np.random.RandomState(2)
    B = rng.randn(E.NB, E.NB)
    A = B @ B.T / E.NB + np.eye(E.NB)
    L, Lout, Xout = E.factor(A)  # raises if a wave reads LDS words another wave wrote since the last barrier
    Lr = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lr)) < 1e-13
    for s in range(E.NT):
        blk = Lr[16 * s:16 * s + 16, 16 * s:16 * s + 16]
        assert np.max(np.abs(E.from_tile(Xout[s]).T - np.linalg.inv(blk))) < 1e-12
    Apan = rng.randn(32, E.NB)
    assert np.max(np.abs(E.panel(Lout, Xout, Apan) - Apan @ np.linalg.inv(Lr).T)) < 1e-12


def test_work_tile_index_is_the_packed_lower_triangle():
    # tsvgp_chol.h: work_tile_index(col, slot) = 8 col - col (col - 1) / 2 + slot enumerates the 36 lower tiles column by column
    idx = [8 * j - j * (j - 1) // 2 + u for j in range(8) for u in range(8 - j)]
    assert idx == list(range(36))
