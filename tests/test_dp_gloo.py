"""CPU tier: the data-parallel step (flat gradient buffer, sum-loss / global-count scaling, one all-reduce,
replicated Adam) on world_size 2 with gloo must equal the single-process step on the union of the shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyModel(torch.nn.Module):
    """Stand-in with the Classify_node call signature (the HIP layers need a GPU; the DP logic does not)."""

    def __init__(self, F, C):
        super().__init__()
        self.conv = torch.nn.ModuleList([torch.nn.Linear(F, 16)])   # named like network.py's modules: the trainer
        self.lt1 = torch.nn.Linear(16, C)                            # all-reduces `conv.0.*` as its own, later bucket

    def forward(self, x, edge_index):
        return torch.log_softmax(self.lt1(torch.tanh(self.conv[0](x))), dim=1)


class Shard:
    def __init__(self, x, y, idx):
        self.x, self.y, self.edge_index = x, y, torch.zeros((2, 0), dtype=torch.long)
        self.train_idx = idx


def _data():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(60, 8, generator=g)
    y = torch.randint(0, 3, (60,), generator=g)
    return x, y


def _worker(rank, world, port, out_q):
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd import train

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data()
    lo, hi = (0, 25) if rank == 0 else (25, 60)  # uneven shards: the global-count scaling matters
    torch.manual_seed(1)
    model = TinyModel(8, 3)
    shard = Shard(x[lo:hi], y[lo:hi], torch.arange(0, hi - lo, 2))
    tr = train.GDTrainer(model, shard, lr=0.01, weight_decay=5e-4)
    assert tr._split > 0   # two buckets: everything above conv.0 is reduced while conv.0's backward still runs
    losses = [float(tr.step()) for _ in range(3)]   # step() returns the GLOBAL loss: it rides in the gradient all-reduce
    out_q.put((rank, losses, {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_single_process():
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd import train

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process on the union
    x, y = _data()
    idx = torch.cat([torch.arange(0, 25, 2), 25 + torch.arange(0, 35, 2)])
    torch.manual_seed(1)
    model = TinyModel(8, 3)
    tr = train.GDTrainer(model, Shard(x, y, idx), lr=0.01, weight_decay=5e-4)
    ref_losses = [float(tr.step()) for _ in range(3)]
    assert np.allclose(res[0][1], ref_losses, rtol=1e-5)
    assert res[0][1] == res[1][1], "every rank reports the same (global) loss"
    for k, v in model.state_dict().items():
        assert np.allclose(res[0][2][k], v.numpy(), rtol=1e-5, atol=1e-6), k
        assert np.array_equal(res[0][2][k], res[1][2][k]), f"ranks diverged on {k}"


def _unseeded_worker(rank, world, port, out_q):
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd import train

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data()
    lo, hi = (0, 25) if rank == 0 else (25, 60)
    torch.manual_seed(100 + 17 * rank)   # what an unseeded launch amounts to: every rank draws its own initial weights
    model = TinyModel(8, 3)
    w_own = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    tr = train.GDTrainer(model, Shard(x[lo:hi], y[lo:hi], torch.arange(0, hi - lo, 2)), lr=0.01, weight_decay=5e-4)
    w_start = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    for _ in range(3):
        tr.step()
    out_q.put((rank, w_own, w_start, {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    torch.distributed.destroy_process_group()


def test_ranks_that_drew_different_weights_train_one_model():
    """--seed defaults to None (main.py:64, as in the reference): each rank then initialises its model from its own generator
    state.  GDTrainer broadcasts rank 0's weights before the first step, so the replicas start -- and, stepping on one
    all-reduced gradient, stay -- identical."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_unseeded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, own0, start0, end0), (_, own1, start1, end1) = res
    assert any(not np.array_equal(own0[k], own1[k]) for k in own0), "the ranks were meant to draw different weights"
    for k in start0:
        assert np.array_equal(start0[k], own0[k]), f"rank 0 keeps its own {k}"
        assert np.array_equal(start1[k], own0[k]), f"rank 1 starts from rank 0's {k}"
        assert np.array_equal(end0[k], end1[k]), f"ranks diverged on {k}"
        assert not np.array_equal(end0[k], start0[k]), f"{k} did not move"


def test_shard_clusters_balances_nnz():
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd.data import shard_clusters

    rng = np.random.default_rng(0)
    nnz = rng.integers(3, 400, size=1000)
    owner = shard_clusters(None, nnz, 8)
    load = np.bincount(owner, weights=nnz, minlength=8)
    assert load.max() - load.min() <= nnz.max()
    assert set(owner.tolist()) == set(range(8))


class TinyGraphModel(torch.nn.Module):
    """Stand-in with the Regress_graph_gc call signature: mean-pool node features per graph, then an MLP."""

    def __init__(self, F):
        super().__init__()
        self.a = torch.nn.Linear(F, 8)
        self.b = torch.nn.Linear(8, 1)

    def forward(self, gc):
        n = gc.num_graphs
        pooled = torch.zeros(n, gc.x.shape[1]).index_add_(0, gc.batch, gc.x)
        cnt = torch.zeros(n).index_add_(0, gc.batch, torch.ones(gc.batch.numel())).clamp(min=1)
        return self.b(torch.tanh(self.a(pooled / cnt.unsqueeze(1))))


def _graph_batches(rank, world):
    """Three global batches of 6, 6 and 3 graphs (4 nodes each); rank r holds graphs r::world of every batch."""
    import types

    g = torch.Generator().manual_seed(3)
    x = torch.randn(15, 4, 5, generator=g)
    y = 5.0 * torch.randn(15, 2, generator=g)
    out = []
    for lo, hi in ((0, 6), (6, 12), (12, 15)):
        ids = list(range(lo, hi))[rank::world]
        if not ids:
            out.append(None)
            continue
        xb = x[ids].reshape(-1, 5)
        bt = torch.repeat_interleave(torch.arange(len(ids)), 4)
        out.append(dict(y=y[ids], n_graphs=len(ids),
                        gc=types.SimpleNamespace(x=xb, edge_index=None, batch=bt, num_graphs=len(ids))))
    return out


def _graph_worker(rank, world, port, out_q):
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd import train

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(2)
    model = TinyGraphModel(5)
    tr = train.GraphTrainer(model, None, None, kind="gc", lr=0.01, weight_decay=5e-4, prop=1,
                            batches=_graph_batches(rank, world), global_sizes=[6, 6, 3])
    losses = [float(tr.step()) for _ in range(3)]
    out_q.put((rank, losses, {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    torch.distributed.destroy_process_group()


def test_graph_level_two_rank_steps_equal_single_process():
    """GraphTrainer data parallel over world_size 2 (gloo): per-batch steps with the reference's never-cleared gradients
    == the single-process trainer on the whole batches (losses and weights); the last batch is short (3 graphs: 2 + 1)."""
    sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
    from fitgnn_amd import train

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(2)
    model = TinyGraphModel(5)
    tr = train.GraphTrainer(model, None, None, kind="gc", lr=0.01, weight_decay=5e-4, prop=1, batches=_graph_batches(0, 1),
                            global_sizes=[6, 6, 3], world=1, rank=0)
    ref = [float(tr.step()) for _ in range(3)]
    assert np.allclose(res[0][1], ref, rtol=1e-5), (res[0][1], ref)
    assert np.allclose(res[1][1], ref, rtol=1e-5)
    for k, v in model.state_dict().items():
        assert np.allclose(res[0][2][k], v.numpy(), rtol=1e-5, atol=1e-6), k
        assert np.array_equal(res[0][2][k], res[1][2][k]), f"ranks diverged on {k}"
