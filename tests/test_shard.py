"""Multi-GPU path on CPU: world_size-2 gloo processes drive the SAME sharding / all-gather
code as the product path, with the oracle injected as the evaluator."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from ccgp_amd import shard


def test_shard_bounds_cover_exactly():
    for total in (0, 1, 7, 64, 512, 624):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == shard.shard_sizes(total, world)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import ccgp_amd  # noqa: F401
    from ccgp_amd import shard as sh
    from conftest import load_maximin
    from oracle import ccgp_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = load_maximin(14)
    y = np.array([orc.test_function_2d(a, b, 3) for a, b in D])
    rng = np.random.default_rng(0)
    B = 13  # not divisible by world: ragged shards
    params = np.stack([orc.params_from_iso(rng.uniform(0.5, 0.95), rng.uniform(0.3, 2), rng.uniform(4, 20), 2)
                       for _ in range(B)])
    calls = []

    def evaluate(ps):
        calls.append(ps.shape[0])
        return np.array([orc.loglik_general(D, y, *orc.unpack_params(r, 2, 2), 0.4)[0] for r in ps])

    ll = sh.sharded_loglik(evaluate, params)

    hyper = np.array([[3, 1, 5, 8], [4, 1.5, 6, 10], [5, 2, 7, 12], [3, 1, 6, 10], [4, 2, 5, 9]], dtype=float)

    def rows(hs):
        return np.array([orc.likeli_hyperpars(D, y, h[:2], h[2:], 0.4, N=16, tau=100.0) for h in hs])

    vals, arg = sh.sharded_grid(rows, hyper)

    Xt = np.array([[0.2, 0.3], [0.7, 0.1], [0.5, 0.5]])

    def pred(ps):
        draws = [(r[0], r[2], r[4]) for r in ps]
        m, v, _ = orc.predict_table(D, y, draws, Xt, 0.4) if len(draws) else (np.empty((0, 3)), np.empty((0, 3)), None)
        return m, v

    mean, var = sh.sharded_predict(pred, params)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), ll=ll, vals=vals, arg=arg, mean=mean, var=var,
             local=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(tmp_path, "r0.npz"))
    r1 = np.load(os.path.join(tmp_path, "r1.npz"))
    for k in ("ll", "vals", "mean", "var"):
        np.testing.assert_array_equal(r0[k], r1[k])       # every rank holds the full result
    assert int(r0["arg"]) == int(r1["arg"])
    assert r0["local"][0] == 7 and r1["local"][0] == 6    # ragged contiguous shards

    # single-process reference: identical per evaluation (evaluations are independent)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_maximin
    from oracle import ccgp_oracle as orc
    D = load_maximin(14)
    y = np.array([orc.test_function_2d(a, b, 3) for a, b in D])
    rng = np.random.default_rng(0)
    params = np.stack([orc.params_from_iso(rng.uniform(0.5, 0.95), rng.uniform(0.3, 2), rng.uniform(4, 20), 2)
                       for _ in range(13)])
    want = np.array([orc.loglik_general(D, y, *orc.unpack_params(r, 2, 2), 0.4)[0] for r in params])
    np.testing.assert_array_equal(r0["ll"], want)
    assert r0["ll"].shape == (13,) and r0["mean"].shape == (13, 3)


def test_gather_buffers_are_staged_on_the_gpu_for_rccl():
    """RCCL ("nccl") moves device memory only: host arrays coming out of the C ABI must be staged through the
    rank's GPU; gloo gathers host memory in place; device tensors stay where they are."""
    import torch
    from ccgp_amd import shard as sh
    cpu = torch.device("cpu")
    assert sh.gather_device("gloo", cpu) == cpu
    cuda1 = torch.device("cuda", 1)
    assert sh.gather_device("nccl", cuda1) == cuda1
    assert sh.gather_device("gloo", cuda1) == cuda1
    if torch.cuda.is_available():
        assert sh.gather_device("nccl", cpu).type == "cuda"


def _gatherer_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import ccgp_amd  # noqa: F401
    from ccgp_amd import shard as sh

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # (a) the likelihood vector of a hyperprior grid: 5 grid rows of 4 nodes over 3 ranks (ragged: 2, 2, 1)
    g = sh.RowGatherer(5, tail=(4,))
    lo, hi = sh.shard_bounds(5, rank, world)
    whole = torch.arange(20, dtype=torch.float64).view(5, 4)
    ptr = (g.send.data_ptr(), g.recv.data_ptr(), g.full.data_ptr())
    for it in range(3):                                   # repeated calls reuse the same three buffers
        out = g.gather(whole[lo:hi] + it)
        assert torch.equal(out, whole + it)
        assert ptr == (g.send.data_ptr(), g.recv.data_ptr(), g.full.data_ptr())
    res["grid"] = out.numpy().copy()
    # (b) evenly divisible and nothing in front of the sharded dimension: the receive buffer is the result
    g2 = sh.RowGatherer(6, tail=(2,))
    lo2, hi2 = sh.shard_bounds(6, rank, world)
    w2 = torch.arange(12, dtype=torch.float64).view(6, 2)
    assert torch.equal(g2.gather(w2[lo2:hi2]), w2) and g2.full.data_ptr() == g2.recv.data_ptr()
    # (c) the (draw x test point) tables of config 5: [2 M, S_local] blocks, draws LAST, 7 draws over 3 ranks; the
    # rank computes straight into the send buffer when its share fills the slot
    M, S = 3, 7
    g3 = sh.RowGatherer(S, lead=(2 * M,))
    lo3, hi3 = sh.shard_bounds(S, rank, world)
    table = torch.arange(2 * M * S, dtype=torch.float64).view(2 * M, S)
    if hi3 - lo3 == g3.slot:
        g3.send.copy_(table[:, lo3:hi3])
        got = g3.gather(None)
    else:
        got = g3.gather(table[:, lo3:hi3].contiguous())
    assert torch.equal(got, table)
    res["tables"] = got.numpy().copy()
    np.savez(os.path.join(out_dir, "g%d.npz" % rank), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_preallocated_gatherer_three_ranks_ragged(tmp_path):
    """shard.RowGatherer (what bench.py's timed step calls): one all_gather_into_tensor into buffers allocated once,
    ragged shards, the sharded dimension first (likelihood vectors) or last (column-major prediction tables)."""
    world = 3
    mp.spawn(_gatherer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(tmp_path, "g%d.npz" % r)) for r in range(world)]
    for k in ("grid", "tables"):
        for o in outs[1:]:
            np.testing.assert_array_equal(o[k], outs[0][k])
