#!/usr/bin/env python3
"""(CPU) Lane-level NumPy emulation of the 128 x 128 diagonal-block Cholesky kernel `potrf_diag2_kernel`
(t-svgp_amd/csrc/tsvgp_chol.hip): every 16 x 16 tile lives in the accumulator layout of v_mfma_f64_16x16x4_f64,
    lane (n = l & 15, G = l >> 4), register r  <->  T[n][4 r + G],
and every step of the factorisation is an MFMA whose operands are registers of other tiles as they stand (see the
kernel's header).  The emulation follows the kernel wave by wave and barrier by barrier: LDS writes of a wave become
visible to the OTHER waves only at the next barrier (a read of a word another wave wrote since the last barrier, or a
word two waves wrote in one interval, raises) -- so it checks the index algebra (slot maps, row ownership, the
progressive look-ahead, the trailing jobs) AND the placement of the barriers.  Run it before touching the kernel.
usage: emul_diag2.py [seed]"""
import sys
import numpy as np

NB, TS = 128, 16
NT = NB // TS  # 8 tile columns; an augmented column has always 9 tiles: (8 - j) of A and (j + 1) of the inverse
NW = 8         # waves: 0 = pivot wave, 1..7 = row owners / trailing workers


def mfma(a, b, c):
    """d = c + A B with A[m][k] = a[m + 16 k], B[k][n] = b[n + 16 k]; d[r][n + 16 G] = C[G + 4 r][n]."""
    A = a.reshape(4, 16).T  # [m][k]
    B = b.reshape(4, 16)    # [k][n]
    C = A @ B
    d = c.copy()
    for r in range(4):
        for G in range(4):
            d[r, 16 * G:16 * G + 16] += C[G + 4 * r, :]
    return d


def to_tile(Tm):
    t = np.zeros((4, 64))
    for r in range(4):
        for G in range(4):
            t[r, 16 * G:16 * G + 16] = Tm[:, 4 * r + G]
    return t


def from_tile(t):
    Tm = np.zeros((16, 16))
    for r in range(4):
        for G in range(4):
            Tm[:, 4 * r + G] = t[r, 16 * G:16 * G + 16]
    return Tm


class Lds:
    """Words keyed by (name, ...) -> 64-lane register image; visibility across waves only through barrier()."""

    def __init__(self):
        self.mem, self.pending = {}, {}

    def write(self, wave, key, val):
        if key in self.pending and self.pending[key][0] != wave:
            raise RuntimeError(f"two waves wrote {key} between two barriers")
        self.pending[key] = (wave, np.array(val, dtype=float).copy())

    def read(self, wave, key):
        if key in self.pending:
            if self.pending[key][0] != wave:
                raise RuntimeError(f"wave {wave} reads {key} written by wave {self.pending[key][0]} since the last barrier")
            return self.pending[key][1].copy()
        return self.mem[key].copy()

    def barrier(self):
        for k, (_, v) in self.pending.items():
            self.mem[k] = v
        self.pending = {}


def pivot_chain(tq, q):
    """4 x 4 pivot chain of group q on the diagonal tile's register q: returns (aop without the rows below, pinv operand)."""
    lane = np.arange(64)
    m_, k_ = lane & 15, lane >> 4
    p = lambda i, j: tq[(4 * q + i) + 16 * j]
    i0 = 1 / np.sqrt(p(0, 0))
    l10, l20, l30 = p(1, 0) * i0, p(2, 0) * i0, p(3, 0) * i0
    d1 = p(1, 1) - l10 * l10
    i1 = 1 / np.sqrt(d1)
    l21, l31 = (p(2, 1) - l20 * l10) * i1, (p(3, 1) - l30 * l10) * i1
    d2 = p(2, 2) - l20 * l20 - l21 * l21
    i2 = 1 / np.sqrt(d2)
    l32 = (p(3, 2) - l30 * l20 - l31 * l21) * i2
    d3 = p(3, 3) - l30 * l30 - l31 * l31 - l32 * l32
    i3 = 1 / np.sqrt(d3)
    R = np.zeros((4, 4))
    R[0, 0], R[1, 1], R[2, 2], R[3, 3] = i0, i1, i2, i3
    R[1, 0] = -l10 * R[0, 0] * R[1, 1]
    R[2, 1] = -l21 * R[1, 1] * R[2, 2]
    R[2, 0] = -(l20 * R[0, 0] + l21 * R[1, 0]) * R[2, 2]
    R[3, 2] = -l32 * R[2, 2] * R[3, 3]
    R[3, 1] = -(l31 * R[1, 1] + l32 * R[2, 1]) * R[3, 3]
    R[3, 0] = -(l30 * R[0, 0] + l31 * R[1, 0] + l32 * R[2, 0]) * R[3, 3]
    Pinv = R.T @ R
    pinv_op = np.where(m_ < 4, Pinv[np.minimum(m_, 3), k_], 0.0)
    aopR0 = np.where((m_ < 4) & (k_ <= m_), R[np.minimum(m_, 3), k_], 0.0)
    shifted = np.zeros(64)  # DPP row_shr:4q
    for l in range(64):
        if (l & 15) >= 4 * q:
            shifted[l] = aopR0[l - 4 * q]
    return shifted, pinv_op


def tile_step(t, q, aop):
    bq = t[q].copy()
    t = t.copy()
    t[q] = 0.0
    return mfma(aop, bq, t)


def row_owner(i):
    """Wave that owns tile row i (1 .. 7): the tile (i, s) of A while s < i.  Row s itself is the pivot wave's."""
    return i


def trailing_jobs(p):
    """Rank-16 update of column p on the columns j >= p + 2, flattened column by column: (j, u) = tile (j + u, j), paired
    with column p slot u + (j - p), A operand = column p slot j - p.  p = -1 enumerates every tile of the columns >= 1
    (the staging jobs of column 0's phase)."""
    return [(j, u) for j in range(p + 2, NT) for u in range(0, NT - j)]


def run_job(lds, w, p, jobs, g):
    """Trailing job g of column p by wave w: tile (j, u) -= sum_kk P_jp[kk] (x) P_(j+u)p[kk] (read-modify-write in LDS)."""
    if p < 0 or g >= len(jobs):
        return
    j, u = jobs[g]
    av = [lds.read(w, ("col", p, j - p, kk)) for kk in range(4)]
    bv = [lds.read(w, ("col", p, u + (j - p), kk)) for kk in range(4)]
    c = np.array([lds.read(w, ("col", j, u, r)) for r in range(4)])
    for kk in range(4):
        c = mfma(-av[kk], bv[kk], c)
    for r in range(4):
        lds.write(w, ("col", j, u, r), c[r])


def factor(Ablk):
    """Tile dataflow of potrf_diag2_kernel.  Returns (L, [X_ss]): the factor and the inverses of its 16 x 16 diagonal tiles."""
    full = np.tril(Ablk) + np.tril(Ablk, -1).T
    lds = Lds()
    lane = np.arange(64)
    m_ = lane & 15
    ident = to_tile(np.eye(16))
    gA = lambda i, j: to_tile(full[16 * i:16 * i + 16, 16 * j:16 * j + 16])
    Lout = {}  # tiles written to global memory: (i, j) -> registers
    Xout = []
    D = gA(0, 0)  # pivot wave: the diagonal tile
    tiles = {w: [] for w in range(1, NW)}
    for i in range(1, NT):
        tiles[row_owner(i)].append(dict(row=i, t=gA(i, 0), la=np.zeros((4, 64))))
    # staging, before the first barrier: every tile of the columns >= 1 from global memory, four per helper (the jobs of p = -1)
    for g, (j, u) in enumerate(trailing_jobs(-1)):
        c = gA(j + u, j)
        for r in range(4):
            lds.write(1 + g % 7, ("col", j, u, r), c[r])
    for s in range(NT):
        jobs = trailing_jobs(s - 1) if s >= 1 else []
        X = ident.copy()  # (rides in wave 7)
        for q in range(4):
            # ---- pivot wave: chain, W, aop, broadcast, own step, the inverse tile's step
            shifted, pinv_op = pivot_chain(D[q], q)
            if q < 3:
                W = mfma(pinv_op, D[q], np.zeros((4, 64)))[0]
                aop = np.where(m_ >= 4 * q + 4, -W, shifted)
            else:
                aop = shifted
            lds.write(0, ("aop", q), aop)
            D = tile_step(D, q, aop)
            lds.barrier()  # G(s, q)
            X = tile_step(X, q, lds.read(NW - 1, ("aop", q)))
            # ---- helpers
            for w in range(1, NW):
                a = lds.read(w, ("aop", q))
                for tl in tiles[w]:
                    u = tl["row"] - s
                    if q >= 1 and s + 1 < NT:  # look-ahead piece kk = q - 1: A operand = register q - 1 of row s + 1's tile
                        a1 = lds.read(w, ("col", s, 1, q - 1))
                        tl["la"] = mfma(-a1, tl["t"][q - 1], tl["la"])
                    tl["t"] = tile_step(tl["t"], q, a)
                    lds.write(w, ("col", s, u, q), tl["t"][q])
                    if q == 3:
                        Lout[(tl["row"], s)] = tl["t"].copy()
                if q < 3:  # one trailing job per barrier interval; the phase's fourth job runs behind barrier E (below)
                    run_job(lds, w, s - 1, jobs, (w - 1) + 7 * q)
        assert len(jobs) <= 28
        Lout[(s, s)] = D.copy()
        Xout.append(X.copy())
        if s + 1 < NT:
            w1 = row_owner(s + 1)
            tl = [x for x in tiles[w1] if x["row"] == s + 1][0]
            tl["la"] = mfma(-tl["t"][3], tl["t"][3], tl["la"])
            nd = np.array([lds.read(w1, ("col", s + 1, 0, r)) for r in range(4)]) + tl["la"]
            for r in range(4):
                lds.write(w1, ("col", s + 1, 0, r), nd[r])
        for r in range(4):
            lds.write(0, ("col", s, 0, r), D[r])  # L_ss: a helper takes it to the matrix behind the barrier
        lds.barrier()  # E(s)
        for w in range(1, NW):
            run_job(lds, w, s - 1, jobs, (w - 1) + 7 * 3)
        if s + 1 < NT:
            D = np.array([lds.read(0, ("col", s + 1, 0, r)) for r in range(4)])
            for w in range(1, NW):
                keep = []
                for tl in tiles[w]:
                    if tl["row"] == s + 1:
                        continue  # became the diagonal tile: this wave only takes jobs from here on
                    a1 = lds.read(w, ("col", s, 1, 3))
                    la = mfma(-a1, tl["t"][3], tl["la"])
                    tl["t"] = np.array([lds.read(w, ("col", s + 1, tl["row"] - (s + 1), r)) for r in range(4)]) + la
                    tl["la"] = np.zeros((4, 64))
                    keep.append(tl)
                tiles[w] = keep
    L = np.zeros((NB, NB))
    for (i, j), t in Lout.items():
        Tm = from_tile(t)
        L[16 * i:16 * i + 16, 16 * j:16 * j + 16] = np.tril(Tm) if i == j else Tm
    # X(s, s)[n][c] = inv(L_ss)[c][n]: as a tile its registers are the A operand of "times inv(L_ss)^T" (see panel())
    return L, Lout, Xout


def panel(Lout, Xout, Apan):
    """chol_panel2_kernel: one wave per 16-row strip of the panel, P = A inv(L_kk)^T by substitution over the eight 16-wide
    column blocks, right-looking: P_s = U_s inv(L_ss)^T, then U_s' -= P_s L_s's^T for s' > s -- all MFMAs on tile registers."""
    n16 = Apan.shape[0] // 16
    P = np.zeros_like(Apan)
    for i in range(n16):
        U = [to_tile(Apan[16 * i:16 * i + 16, 16 * s:16 * s + 16]) for s in range(NT)]
        for s in range(NT):
            # inv(L_ss)[c][k] as a tile with rows c: registers kk hold the columns 4 kk + G.  The pivot wave's X tile holds
            # X[n][c] = inv(L_ss)^T[n][c] = inv(L_ss)[c][n]: its TRANSPOSE is the operand -- the kernel stores it transposed.
            Xs = to_tile(from_tile(Xout[s]).T)
            ps = np.zeros((4, 64))
            for kk in range(4):
                ps = mfma(Xs[kk], U[s][kk], ps)
            P[16 * i:16 * i + 16, 16 * s:16 * s + 16] = from_tile(ps)
            for s2 in range(s + 1, NT):
                Lt = Lout[(s2, s)]
                for kk in range(4):
                    U[s2] = mfma(-Lt[kk], ps[kk], U[s2])
    return P


if __name__ == "__main__":
    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    B = rng.randn(NB, NB)
    A = B @ B.T / NB + np.eye(NB)
    L, Lout, Xout = factor(A)
    Lr = np.linalg.cholesky(A)
    print("max |L - chol(A)|                 ", np.max(np.abs(L - Lr)))
    ex = max(np.max(np.abs(from_tile(Xout[s]).T - np.linalg.inv(Lr[16 * s:16 * s + 16, 16 * s:16 * s + 16]))) for s in range(NT))
    print("max |X_ss^T - inv(L_ss)|          ", ex)
    Apan = rng.randn(48, NB)
    P = panel(Lout, Xout, Apan)
    ep = np.max(np.abs(P - Apan @ np.linalg.inv(Lr).T))
    print("max |panel - A inv(L)^T|          ", ep)
    assert np.max(np.abs(L - Lr)) < 1e-12 and ex < 1e-12 and ep < 1e-11
    print("ok")
