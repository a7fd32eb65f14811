"""GPU tier: BASELINE.json's configurations at their full sizes through the path bench.py times (GDTrainer: de-duplicated
layer-0 table with the direct-gather SpMM, linked epilogues, fused head, fused loss, hidden 512).

The oracle cannot run the whole union (8.2 M rows at S-products), so each configuration is checked in two links:
  1. sampled loader slices -- a few of the reference's 128-subgraph loader batches (run.py:336) cut out of the union
     (data.select_clusters) and run BOTH through the fast path on the GPU and through the torch-CPU oracle: logits
     rows <= 1e-4 relative (north_star's tolerance), loss, and the slices' gradient contribution <= 1e-3;
  2. the full union through the same fast path -- its logits on the sampled slices' rows equal the slice runs
     (subgraphs share no edges, utils.py:248: block-diagonal independence), and its gradient equals the sum of the
     gradients of its shards (data.shard_clusters, the data-parallel partition): additivity over whole subgraphs, the
     identity the data-parallel step rests on (run.py:184-204: one loss over all batches).
Plus one training-mode step (hashed dropout, Adam): finite loss, weights move, a second trainer on the same seeds
reproduces it bit for bit."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = ["S-pubmed", "S-physics", "S-products"]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module", params=CONFIGS)
def cfg(request):
    from fitgnn_amd import workloads

    name = request.param
    dev = torch.device("cuda")
    wl = workloads.coarsen_workload(name, dev, spectral="device")
    ei_d, assign_d = torch.from_numpy(wl["ei"]).to(dev), torch.from_numpy(wl["assign"]).to(dev)
    sub, nnz_c = workloads.assemble(name, ei_d, assign_d, wl["n_clusters"])
    X, y = workloads.features_and_labels(name)
    return dict(name=name, wl=wl, sub=sub, nnz_c=nnz_c, X=X, y=y, dev=dev)


def _model(name, seed=2, dropout=0.5):
    from fitgnn_amd import network, workloads

    N, E, F, C, r = workloads.SHAPES[name]
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=512, num_classes=C, dropout=dropout)
    torch.manual_seed(seed)
    return network.Classify_node(args).cuda()


def _fast_path(model, batch, scale):
    """Eval-mode forward + fused loss + backward exactly as GDTrainer.step issues them (dedup table, fused head)."""
    from fitgnn_amd.ops import SoftmaxNLL

    model.eval()
    model.zero_grad()
    z = model.embed_and_head(batch.x_table, batch.edge_index, batch.row_index)
    loss = SoftmaxNLL.apply(z, batch.train_idx, batch.y.index_select(0, batch.train_idx), scale)
    loss.backward()
    return z.detach(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def test_partition_properties(cfg):
    """SURVEY §8c known-answer properties at full size: one cluster per node, cluster count ceil((1 - r) N) per level
    schedule, every node the core of exactly one subgraph, nnz' additive over subgraphs."""
    from fitgnn_amd import workloads

    N, E, F, C, r = workloads.SHAPES[cfg["name"]]
    assign, n = cfg["wl"]["assign"], cfg["wl"]["n_clusters"]
    assert assign.shape == (N,) and assign.min() == 0 and assign.max() == n - 1
    assert len(np.unique(assign)) == n
    assert n == int(np.ceil((1 - r) * N)), "the greedy selection reaches the target size on these graphs"
    Cc = cfg["wl"]["C"]
    assert np.allclose(np.asarray(Cc.power(2).sum(1)).ravel(), 1.0, rtol=0, atol=1e-12)
    sub = cfg["sub"]
    assert int(sub["core"].sum()) == N
    assert int(cfg["nnz_c"].sum()) == int(sub["edge_index"].shape[1]) + int(sub["ptr"][-1])


def test_sampled_loader_slices_match_the_oracle_and_the_full_union(cfg):
    from fitgnn_amd import data, workloads
    from oracle import gnn_oracle as gorc

    name, sub = cfg["name"], cfg["sub"]
    n_c = cfg["wl"]["n_clusters"]
    model = _model(name)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    full = workloads.batch_from_subgraphs(name, sub, cfg["dev"], cfg["X"], cfg["y"])
    scale = 1.0 / float(full.train_idx.numel())
    z_full, loss_full, g_full = _fast_path(model, full, scale)
    assert np.isfinite(loss_full)
    n_batches = (n_c + 127) // 128
    picks = sorted({0, n_batches // 3, (2 * n_batches) // 3, n_batches - 1})
    ptr = sub["ptr"].cpu().numpy()
    for b in picks:
        clusters = np.arange(b * 128, min((b + 1) * 128, n_c))
        part = workloads.batch_from_subgraphs(name, data.select_clusters(sub, clusters), cfg["dev"], cfg["X"], cfg["y"])
        z, loss, g = _fast_path(model, part, scale)
        r0, r1 = int(ptr[clusters[0]]), int(ptr[clusters[-1] + 1])
        assert part.n_rows == r1 - r0
        # link 2: the full union's rows of this slice == the slice run
        assert rel(z_full[r0:r1].cpu(), z.cpu()) < 2e-5, (name, b)
        # link 1: the slice run == the oracle
        o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, part.x.cpu(), part.edge_index.cpu(), part.y.cpu(), num_layers=2,
                                                         train_mask=part.train_mask.cpu(), loss_scale=scale)
        assert rel(torch.log_softmax(z, 1).cpu(), o_ref) < 1e-4, (name, b)   # Classify_node's output (network.py:35)
        assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (name, b, loss, float(l_ref))
        for k in g:
            assert rel(g[k].cpu(), g_ref[k]) < 1e-3, (name, b, k)
        del part


def test_gradient_is_additive_over_shards(cfg):
    """Full-size identity behind the data-parallel step: with the global 1 / count scale, the gradient of the whole union
    equals the sum of the gradients of its data.shard_clusters() shards."""
    from fitgnn_amd import data, workloads

    name, sub = cfg["name"], cfg["sub"]
    model = _model(name)
    full = workloads.batch_from_subgraphs(name, sub, cfg["dev"], cfg["X"], cfg["y"])
    scale = 1.0 / float(full.train_idx.numel())
    _, loss_full, g_full = _fast_path(model, full, scale)
    del full
    world = 3
    owner = data.shard_clusters(None, cfg["nnz_c"], world)
    load = np.bincount(owner, weights=cfg["nnz_c"], minlength=world)
    assert load.max() - load.min() <= cfg["nnz_c"].max()
    acc, loss_sum = None, 0.0
    for k in range(world):
        part = workloads.batch_from_subgraphs(name, data.select_clusters(sub, np.nonzero(owner == k)[0]), cfg["dev"], cfg["X"], cfg["y"])
        _, loss, g = _fast_path(model, part, scale)
        loss_sum += loss
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
        del part
    assert abs(loss_sum - loss_full) <= 1e-5 * abs(loss_full)
    for n in g_full:
        assert rel(acc[n], g_full[n]) < 2e-4, n


def test_training_step_is_finite_moves_the_weights_and_reproduces(cfg):
    from fitgnn_amd import train, workloads

    name = cfg["name"]
    batch = workloads.batch_from_subgraphs(name, cfg["sub"], cfg["dev"], cfg["X"], cfg["y"])
    runs = []
    for _ in range(2):
        model = _model(name)
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        torch.manual_seed(11)   # dropout seeds are drawn from torch's generator
        tr = train.GDTrainer(model, batch, lr=0.01, weight_decay=5e-4)
        losses = [float(tr.step()) for _ in range(2)]
        assert all(np.isfinite(losses))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        assert any(not torch.equal(sd[k], w0[k]) for k in sd)
        runs.append((losses, sd))
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
