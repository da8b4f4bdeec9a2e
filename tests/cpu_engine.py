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
        k = O.SquaredExponential(variance=float(kernel.variance.value), lengthscales=kernel.lengthscales.numpy())
        return torch.as_tensor(k.K(Z.cpu().numpy()))

    def run(self, X, Y, Z, kernel, *, moment_Tm, moment_mode, gamma, lik_id=0, lik_param=0.0, whiten_T=None,
            whiten_mode=1, sites=False, want_moments=False, want_grads=False, b_tag=None):
        k = O.SquaredExponential(variance=float(kernel.variance.value), lengthscales=kernel.lengthscales.numpy())
        Xn, Zn = X.cpu().numpy(), Z.cpu().numpy()
        A = k.K(Xn, Zn)
        tri = {0: np.tril, 1: np.triu, 2: lambda a: a}  # the kernels only read the triangle the mode names
        if whiten_T is not None:
            A = A @ tri[whiten_mode](whiten_T.cpu().numpy()).T
        Tm = tri[moment_mode](moment_Tm.cpu().numpy())
        C = np.einsum("nj,pij->pni", A, Tm)
        q = np.sum(C * C, axis=-1).T
        mean = A @ gamma.cpu().numpy()
        var = k.variance - q
        st = _Stats()
        st.n_rows = Xn.shape[0]
        st.nonpos = torch.tensor(float(np.sum(~(var > 0))))
        st.ve_sum = torch.tensor(0.0, dtype=torch.float64)
        st.acc2 = st.acc1 = st.mean = st.var = st.g0 = st.g1 = None
        if want_moments:
            st.mean, st.var = torch.as_tensor(mean), torch.as_tensor(var)
        if lik_id != 0:
            lik = O.Gaussian(variance=lik_param) if lik_id == 1 else O.Bernoulli()
            Yn = Y.cpu().numpy()
            g0, g1 = lik.variational_expectations_grads(mean, var, Yn)
            g1 = np.minimum(g1, -1e-8)
            st.ve_sum = torch.tensor(float(np.sum(lik.variational_expectations(mean, var, Yn))), dtype=torch.float64)
            if sites:
                st.acc2 = torch.as_tensor(np.einsum("nm,no,nl->lmo", A, A, g1))
                st.acc1 = torch.as_tensor(np.einsum("nm,nl->lm", A, g0))
        return st
