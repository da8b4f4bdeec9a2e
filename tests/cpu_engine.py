"""TEST DOUBLE for ``EStepEngine`` (test infrastructure, like oracle/): a NumPy implementation of the per-shard N-pass
with the same interface, so that the HOST logic of the product -- M x M prelude, packing, the all-reduce over ranks and
the site-update epilogue -- can be exercised on CPU (``gloo``, world_size 2) where no HIP kernel can run.
It is never importable from the package and never used on a GPU box."""
import numpy as np
import torch

from oracle import tsvgp_oracle as O


class _Stats:
    pass


class NumpyShardEngine:
    def kuu(self, Z, kernel):
        if hasattr(kernel, "kernels"):  # SeparateIndependent: [P, M, M]
            return torch.stack([self.kuu(Z, k) for k in kernel.kernels])
        k = getattr(O, type(kernel).__name__)(variance=float(kernel.variance.value), lengthscales=kernel.lengthscales.numpy())
        return torch.as_tensor(k.K(Z.cpu().numpy()))

    def run(self, X, Y, Z, kernel, *, moment_Tm, moment_mode, gamma, lik_id=0, lik_param=0.0, whiten_T=None,
            whiten_mode=1, project_T=None, sites=False, want_moments=False, want_grads=False, b_tag=None,
            mean_only=False, prefill=None, moments_on_kfu=False, project_mode=0):
        if hasattr(kernel, "kernels"):  # one pass per latent, as EStepEngine._run_separate
            parts = [self.run(X, None if Y is None else Y[:, p:p + 1], Z, kp, moment_Tm=moment_Tm[p:p + 1],
                              moment_mode=moment_mode, gamma=gamma[:, p:p + 1], lik_id=lik_id, lik_param=lik_param,
                              whiten_T=(whiten_T[p] if isinstance(whiten_T, (list, tuple)) else None if whiten_T is None
                                        else (whiten_T[p] if whiten_T.dim() == 3 else whiten_T)),
                              whiten_mode=whiten_mode, sites=sites, want_moments=want_moments, want_grads=want_grads,
                              project_T=(project_T[p] if isinstance(project_T, (list, tuple)) else None if project_T is None
                                         else (project_T[p] if project_T.dim() == 3 else project_T)),
                              mean_only=mean_only, moments_on_kfu=moments_on_kfu, project_mode=project_mode)
                     for p, kp in enumerate(kernel.kernels)]
            st = _Stats()
            st.n_rows = parts[0].n_rows
            st.nonpos, st.ve_sum = sum(s.nonpos for s in parts), sum(s.ve_sum for s in parts)
            for name, dim in (("mean", 1), ("var", 1), ("g0", 1), ("g1", 1), ("acc2", 0), ("acc1", 0)):
                vals = [getattr(s, name) for s in parts]
                setattr(st, name, None if vals[0] is None else torch.cat(vals, dim=dim))
            return st
        k = getattr(O, type(kernel).__name__)(variance=float(kernel.variance.value), lengthscales=kernel.lengthscales.numpy())
        Xn, Zn = X.cpu().numpy(), Z.cpu().numpy()
        A = k.K(Xn, Zn)
        tri = {0: np.tril, 1: np.triu, 2: lambda a: a}  # the kernels only read the triangle the mode names
        Aw = A if whiten_T is None else A @ tri[whiten_mode](whiten_T.cpu().numpy()).T
        Am = A if moments_on_kfu else Aw  # operand of the moments
        A = Aw  # operand of the site sums
        Tm = tri[moment_mode](moment_Tm.cpu().numpy())
        C = np.einsum("nj,pij->pni", Am, Tm)
        q = np.sum(C * C, axis=-1).T
        mean = Am @ gamma.cpu().numpy()
        var = k.variance - q
        st = _Stats()
        st.n_rows = Xn.shape[0]
        st.nonpos = torch.tensor(float(np.sum(~(var > 0))))
        st.ve_sum = torch.tensor(0.0, dtype=torch.float64)
        st.acc2 = st.acc1 = st.mean = st.var = st.g0 = st.g1 = None
        if want_moments:
            st.mean, st.var = torch.as_tensor(mean), torch.as_tensor(var)
        crop, lik_id = not (lik_id & 0x100), lik_id & 0xFF  # TSVGP_LIK_NOCROP
        if lik_id != 0:
            lik = O.Gaussian(variance=lik_param) if lik_id == 1 else O.Bernoulli()
            Yn = Y.cpu().numpy()
            g0, g1 = lik.variational_expectations_grads(mean, var, Yn)
            g1 = np.minimum(g1, -1e-8) if crop else g1
            st.ve_sum = torch.tensor(float(np.sum(lik.variational_expectations(mean, var, Yn))), dtype=torch.float64)
            if sites:
                As = A if project_T is None else A @ tri[project_mode](project_T.cpu().numpy()).T  # projected route: a = K9^-1 k
                st.acc2 = torch.as_tensor(np.einsum("nm,no,nl->lmo", As, As, g1))
                st.acc1 = torch.as_tensor(np.einsum("nm,nl->lm", As, g0))
        if mean_only:  # TSVGP_LIK_MEANONLY: no variance, no variational expectation; non-finite rows are counted
            assert lik_id in (0, 1)
            st.var, st.ve_sum = None, torch.tensor(float("nan"), dtype=torch.float64)
            st.nonpos = torch.tensor(float(np.sum(~np.isfinite(mean))))
        return st
