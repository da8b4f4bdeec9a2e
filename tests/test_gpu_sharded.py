"""The N-sharded E-step on the GPU with more than one process: two ranks share cuda:0 (this box has one GPU) and
all-reduce over ``gloo``; on an 8-GPU node the same code runs one rank per GPU over RCCL (bench.py --gpus N).
Checks the sharded HIP path end to end against the single-process HIP path and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tsvgp_oracle as O
from tests.helpers import free_port, pkg, relerr, synthetic

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        X, Y, Z = synthetic(N=5001, M=96, D=4, lik="bernoulli", seed=9)
        Xs, Ys = p.distributed.shard_rows(X, Y)
        m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=5001, device="cuda:0")
        Xd, Yd = torch.as_tensor(Xs, device="cuda:0"), torch.as_tensor(Ys, device="cuda:0")
        for _ in range(3):
            m.natgrad_step((Xd, Yd), lr=0.8)
        e = float(m.elbo((Xd, Yd)))
        e2, grads = m.elbo_and_grads((Xd, Yd))  # M-step gradient: partial sums all-reduced like the accumulators
        if rank == 0:
            np.savez(out, l1=m.lambda_1.numpy(), L2=m.lambda_2.cpu().numpy(), elbo=e, elbo2=float(e2),
                     g_var=grads["variance"].cpu().numpy(), g_ls=grads["lengthscales"].cpu().numpy(), g_Z=grads["Z"].cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=5001, M=96, D=4, lik="bernoulli", seed=9)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=5001)
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-8
    assert relerr(got["L2"], ora.lambda_2) < 1e-8
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    # sharded gradients == single-process gradients (same state: load the oracle's, identical to 1e-8)
    p = pkg()
    single = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=5001, device="cuda:0",
                      lambda_1=got["l1"], lambda_2_sqrt=np.linalg.cholesky(got["L2"]) * -1.0)
    e1, g1 = single.elbo_and_grads((X, Y))
    assert abs(float(got["elbo2"]) - float(e1)) < 1e-10 * abs(float(e1))
    assert relerr(got["g_var"], g1["variance"].cpu().numpy()) < 1e-8
    assert relerr(got["g_ls"], g1["lengthscales"].cpu().numpy()) < 1e-8
    assert relerr(got["g_Z"], g1["Z"].cpu().numpy()) < 1e-7


def _worker_empty(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        X, Y, Z = synthetic(N=2, M=2, D=2, seed=3)
        X, Y = X[:1], Y[:1]  # one row for two ranks: rank 1 holds an empty shard
        Xs, Ys = p.distributed.shard_rows(X, Y)
        assert Xs.shape[0] == (1 if rank == 0 else 0)
        m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, num_data=1, device="cuda:0")
        Xd, Yd = torch.as_tensor(Xs, device="cuda:0"), torch.as_tensor(Ys, device="cuda:0")
        m.natgrad_step((Xd, Yd), lr=0.8)
        e = float(m.elbo((Xd, Yd)))
        if rank == 0:
            np.savez(out, l1=m.lambda_1.numpy(), L2=m.lambda_2.cpu().numpy(), elbo=e)
    finally:
        dist.destroy_process_group()


def test_empty_shard_contributes_zeros(tmp_path):
    """N < world size: the rank without rows takes part in the all-reduce with zeros instead of raising alone while the
    others wait (``shard_bounds`` hands it an empty block); a single process with no rows still raises."""
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker_empty, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=2, M=2, D=2, seed=3)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z, num_data=1)
    ora.natgrad_step((X[:1], Y[:1]), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-8 and relerr(got["L2"], ora.lambda_2) < 1e-8
    assert abs(float(got["elbo"]) - ora.elbo((X[:1], Y[:1]))) < 1e-9 * abs(ora.elbo((X[:1], Y[:1])))
    p = pkg()
    m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, device="cuda:0")
    with pytest.raises(ValueError):
        m.natgrad_step((X[:0], Y[:0]), lr=0.8)


def _worker_graph(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        X, Y, Z = synthetic(N=3001, M=64, D=4, lik="bernoulli", seed=10)
        Xs, Ys = p.distributed.shard_rows(X, Y)
        m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=3001, device="cuda:0", use_graph=True)
        m.GRAPH_FORK_MIN_NM = 0  # fork the fill inside the capture, as a shard of N * M >= 1e8 does (one rank's share of 8)
        Xd, Yd = torch.as_tensor(Xs, device="cuda:0"), torch.as_tensor(Ys, device="cuda:0")
        for _ in range(5):  # eager, capture + replay, three more replays
            m.natgrad_step((Xd, Yd), lr=0.8)
        captured = any(isinstance(e, dict) and "tail" in e for e in m._graphs.values())
        e = float(m.elbo((Xd, Yd)))
        if rank == 0:
            np.savez(out, l1=m.lambda_1.numpy(), L2=m.lambda_2.cpu().numpy(), elbo=e, captured=captured)
    finally:
        dist.destroy_process_group()


def test_two_rank_step_replayed_as_two_graphs_around_the_all_reduce(tmp_path):
    """use_graph with more than one rank: the step is captured as two graphs (in front of and behind the all-reduce of
    the packed accumulators) and replayed with the collective issued in between; results against the oracle."""
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker_graph, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert bool(got["captured"])
    X, Y, Z = synthetic(N=3001, M=64, D=4, lik="bernoulli", seed=10)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=3001)
    for _ in range(5):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-8
    assert relerr(got["L2"], ora.lambda_2) < 1e-8
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))


def _worker_rccl(rank, world, port, out):
    """ONE rank on backend "nccl" (RCCL) with the collectives forced on: the calls an 8-GPU run makes -- group initialisation
    with a device id, all-reduce of the packed accumulators between the N-pass and the epilogue on the step's stream, the
    broadcast of the route decision, the step captured as two hipGraphs with the collective between the replays -- on the one
    GPU of this box."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        p = pkg()
        p.distributed.FORCE_COLLECTIVES = True
        assert p.distributed.collectives_on()
        calls = {"all_reduce": 0, "broadcast": 0}
        real_ar, real_bc = dist.all_reduce, dist.broadcast

        def counted_ar(t, *a, **k):
            assert t.is_cuda  # the RCCL path: device buffers, no host staging
            calls["all_reduce"] += 1
            return real_ar(t, *a, **k)

        def counted_bc(t, *a, **k):
            calls["broadcast"] += 1
            return real_bc(t, *a, **k)

        dist.all_reduce, dist.broadcast = counted_ar, counted_bc
        X, Y, Z = synthetic(N=4001, M=160, D=4, lik="bernoulli", seed=12)
        Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
        res = {}
        for tag, use_graph in (("eager", False), ("graph", True), ("graphfork", True)):
            m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=4001, device="cuda:0", use_graph=use_graph)
            if tag == "graphfork":  # the capture of a large shard: fill and epilogue operands forked onto the side stream
                m.GRAPH_FORK_MIN_NM = 0
            assert m._reduce()
            n0 = calls["all_reduce"]
            for _ in range(5):  # graph: eager, capture + replay, three more replays
                m.natgrad_step((Xd, Yd), lr=0.8)
            res[tag + "_reduces"] = calls["all_reduce"] - n0
            if use_graph:
                res["captured"] = any(isinstance(e, dict) and "tail" in e for e in m._graphs.values())
            res[tag + "_l1"], res[tag + "_L2"] = m.lambda_1.numpy(), m.lambda_2.cpu().numpy()
            res[tag + "_elbo"] = float(m.elbo((Xd, Yd)))
            e2, grads = m.elbo_and_grads((Xd, Yd))
            res[tag + "_elbo2"] = float(e2)
        res["broadcasts"] = calls["broadcast"]
        np.savez(out, **res)
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_drives_the_collective_path(tmp_path):
    """Backend nccl = RCCL, world_size 1, collectives forced on (``distributed.FORCE_COLLECTIVES``): eager steps, the
    two-graph replay around the all-reduce, the route broadcast and the ELBO / gradient reductions all go through RCCL on
    device buffers and must reproduce the oracle (reference src/models/tsvgp.py:278-281, :95 summed over ranks)."""
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker_rccl, args=(1, port, out), nprocs=1, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=4001, M=160, D=4, lik="bernoulli", seed=12)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=4001)
    for _ in range(5):
        ora.natgrad_step((X, Y), lr=0.8)
    e = ora.elbo((X, Y))
    assert bool(got["captured"])
    assert int(got["eager_reduces"]) == 5 and int(got["graph_reduces"]) == 5  # one all-reduce per step, replayed or not
    assert int(got["graphfork_reduces"]) == 5
    assert int(got["broadcasts"]) >= 1  # the route decision (cond(K_uu + jitter I)) came from rank 0
    for tag in ("eager", "graph", "graphfork"):
        assert relerr(got[tag + "_l1"], ora.lambda_1) < 1e-8
        assert relerr(got[tag + "_L2"], ora.lambda_2) < 1e-8
        assert abs(float(got[tag + "_elbo"]) - e) < 1e-9 * abs(e)
        assert abs(float(got[tag + "_elbo2"]) - e) < 1e-9 * abs(e)


def _worker_split(rank, world, port, out, backend):
    """One kernel per latent on several ranks: the latents' M x M algebra split over the ranks (t_SVGP._step_device_split)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        if world == 1:
            p.distributed.FORCE_COLLECTIVES = True
        P = 3
        X, Y, Z = synthetic(N=3001, M=130, D=5, P=P, lik="gaussian", seed=13)
        ls = (0.9, 1.6, 1.2)  # cond(K_uu + 1e-9 I) = 9e2, 1.8e5, 1.1e4: the middle latent is beyond the direct route's gate
        kern = lambda mod: mod.SeparateIndependent([mod.SquaredExponential(1.0, l) for l in ls])
        m = p.t_SVGP(kern(p), p.Gaussian(0.1), p.SharedIndependentInducingVariables(Z), num_latent_gps=P, num_data=3001,
                     device="cuda:0")
        Xs, Ys = p.distributed.shard_rows(X, Y)
        Xd, Yd = torch.as_tensor(Xs, device="cuda:0"), torch.as_tensor(Ys, device="cuda:0")
        routes = m._routes(1e-9)
        assert routes == ["direct", "whitened", "direct"] and m._latent_split(routes)
        for _ in range(3):
            m.natgrad_step((Xd, Yd), lr=0.8)
        e = float(m.elbo((Xd, Yd)))
        if rank == 0:
            np.savez(out, l1=m.lambda_1.numpy(), L2=m.lambda_2.cpu().numpy(), elbo=e, routes=np.array(routes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_latent_split_over_ranks_matches_oracle(tmp_path, world, backend):
    """BASELINE configs[4] across GPUs: every latent's prelude / epilogue has one owner, the N-pass operands
    are all-gathered, the accumulators reduce-scattered by latent, the new state all-gathered.  Two ranks on the one GPU
    over gloo, and one rank over RCCL with the collectives forced (all_gather_into_tensor / reduce_scatter_tensor on device
    buffers); against the oracle (reference src/models/tsvgp.py:249-254, 268-303 with K_uu [P, M, M])."""
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker_split, args=(world, port, out, backend), nprocs=world, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=3001, M=130, D=5, P=3, lik="gaussian", seed=13)
    ora = O.t_SVGP(O.SeparateIndependent([O.SquaredExponential(1.0, l) for l in (0.9, 1.6, 1.2)]), O.Gaussian(0.1),
                   O.SharedIndependentInducingVariables(Z), num_latent_gps=3, num_data=3001)
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-8
    assert relerr(got["L2"], ora.lambda_2) < 1e-8
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] ("C4"): configs[2] -- Bernoulli probit, M = 1024, D = 16, fp32 N-arrays -- sharded over the ranks
# ---------------------------------------------------------------------------------------------------------------------
C4_ROWS = 6001  # 47 row panels, the last one ragged; uneven shards (3001 + 3000); the fp64 oracle takes ~2 s per step here
C4_STEPS = 4  # eager, capture + replay, two more replays


def _c4_problem():
    import bench

    w = dict(bench.WORKLOADS["c3"], N=C4_ROWS)
    assert (w["M"], w["D"], w["lik"], w["dtype"]) == (1024, 16, "bernoulli", "f32")
    return bench.make_data(w)  # bench.py's own generator: what `bench.py --workload c3 --gpus N` shards


def _worker_c4(rank, world, port, out, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        if world == 1:
            p.distributed.FORCE_COLLECTIVES = True
        reduces = {"n": 0}
        real_ar = dist.all_reduce

        def counted(t, *a, **k):
            if t.numel() > 1000:  # the packed accumulators (M (M + 1) / 2 + M + 3 doubles), not the small bookkeeping ones
                assert t.dtype == torch.float64 and t.numel() == 1024 * 1025 // 2 + 1024 + 3
                reduces["n"] += 1
            return real_ar(t, *a, **k)

        dist.all_reduce = counted
        X, Y, Z = _c4_problem()
        Xs, Ys = p.distributed.shard_rows(X, Y)
        Xd = torch.as_tensor(Xs, dtype=torch.float32, device="cuda:0")
        Yd = torch.as_tensor(Ys, dtype=torch.float32, device="cuda:0")
        res = {}
        for tag, use_graph, fork_min in (("eager", False, None), ("graph", True, None), ("graphfork", True, 0)):
            m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=C4_ROWS, device="cuda:0",
                         compute_dtype=torch.float32, use_graph=use_graph)
            if fork_min is not None:  # the capture of a 125 000-row shard: fill and epilogue operands forked onto the side stream
                m.GRAPH_FORK_MIN_NM = fork_min
            assert m._reduce() and m._routes(1e-9) == ["direct"]  # D = 16: cond(K_uu) ~ 3, inside the fp32 gate (30)
            n0 = reduces["n"]
            for _ in range(C4_STEPS):
                m.natgrad_step((Xd, Yd), lr=0.8)
            res[tag + "_reduces"] = reduces["n"] - n0
            if use_graph:
                res[tag + "_captured"] = any(isinstance(e, dict) and "tail" in e for e in m._graphs.values())
            res[tag + "_l1"], res[tag + "_L2"] = m.lambda_1.numpy(), m.lambda_2.cpu().numpy()
            res[tag + "_elbo"] = float(m.elbo((Xd, Yd)))  # all-reduced over the ranks
            m.data_parallel = False  # predictions of the first rows of THIS rank's shard: no collective
            mu, var = m.predict_f(Xd[:300])
            res[tag + "_mu"], res[tag + "_var"] = mu.cpu().numpy(), var.cpu().numpy()
        if rank == 0:
            np.savez(out, **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_c4_sharded_fp32_bernoulli_matches_oracle(tmp_path, world, backend):
    """BASELINE configs[3] as far as one GPU allows: bench.py's `c3` problem (Bernoulli probit GH-20, M = 1024, D = 16, fp32
    N-arrays, fp64 M x M algebra) at N = 6001, row-sharded over TWO ranks sharing cuda:0 (gloo) and over ONE rank on backend
    nccl = RCCL with the collectives forced on; eager steps, the two-graph replay around the all-reduce, and the replay with the
    fill forked inside the capture (what a 125 000-row shard of the 8-GPU run takes).  One all-reduce of the packed fp64
    accumulators per step (reference src/models/tsvgp.py:278-281, :95 summed over the ranks).  Against the fp64 oracle's
    single-process steps at SURVEY 8(d)'s fp32 tolerances: moments atol 1e-4 + rtol 1e-3, |dELBO| / |ELBO| <= 1e-4; the three
    execution modes against each other to summation order."""
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker_c4, args=(world, free_port(), out, backend), nprocs=world, join=True)
    got = np.load(out)
    X, Y, Z = _c4_problem()
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=C4_ROWS)
    for _ in range(C4_STEPS):
        ora.natgrad_step((X, Y), lr=0.8)
    e_o = ora.elbo((X, Y))
    mu_o, var_o = ora.predict_f(X[:300])  # rank 0's shard starts at row 0
    assert bool(got["graph_captured"]) and bool(got["graphfork_captured"])
    for tag in ("eager", "graph", "graphfork"):
        assert int(got[tag + "_reduces"]) == C4_STEPS, tag  # ONE packed all-reduce per step, replayed or not
        assert abs(float(got[tag + "_elbo"]) - e_o) < 1e-4 * abs(e_o), tag
        np.testing.assert_allclose(got[tag + "_mu"], mu_o, rtol=1e-3, atol=1e-4, err_msg=tag)
        np.testing.assert_allclose(got[tag + "_var"], var_o, rtol=1e-3, atol=1e-4, err_msg=tag)
        # the site parameters themselves: fp32 rounding of K(X, Z) enters the sums of 6001 rows at ~1e-5 relative
        assert relerr(got[tag + "_l1"], ora.lambda_1) < 2e-3, tag
        assert relerr(got[tag + "_L2"], ora.lambda_2) < 2e-3, tag
        # replayed == eager: same kernels, same shards, same summation order
        assert relerr(got[tag + "_l1"], got["eager_l1"]) < 1e-10 and relerr(got[tag + "_L2"], got["eager_L2"]) < 1e-10, tag
