"""GPU tier: PyG-signature layers and the network.py classes vs the torch-CPU oracle (eval mode and
injected-dropout training mode), logits <= 1e-4 rel, gradients <= 1e-3 rel."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from fitgnn_amd import network
    from fitgnn_amd import nn as fnn
    from oracle import gnn_oracle as gorc

    return network, fnn, gorc


def _lib_mod():
    from fitgnn_amd import _lib
    return _lib


def graph(n=300, m=900, seed=0):
    rng = np.random.default_rng(seed)
    a, b = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
    k = a != b
    ei = np.unique(np.concatenate([np.stack([a[k], b[k]]), np.stack([b[k], a[k]])], 1), axis=1)
    return torch.tensor(ei, dtype=torch.long), n


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))


def test_gcnconv_forward_backward(mods):
    network, fnn, gorc = mods
    ei, n = graph()
    torch.manual_seed(0)
    conv = fnn.GCNConv(40, 96).cuda()
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 40)
    xg = x.cuda().requires_grad_(True)
    out = conv(xg, ei.cuda())
    W, b = conv.lin.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = gorc.gcn_conv(xc, ei, W, b)
    assert rel(out.detach().cpu(), ref.detach()) < 1e-4
    gout = torch.randn(n, 96)
    out.backward(gout.cuda())
    ref.backward(gout)
    assert rel(xg.grad.cpu(), xc.grad) < 1e-4
    assert rel(conv.lin.weight.grad.cpu(), W.grad) < 1e-3
    assert rel(conv.bias.grad.cpu(), b.grad) < 1e-4


def test_sage_gin_appnp(mods):
    network, fnn, gorc = mods
    ei, n = graph(seed=3)
    torch.manual_seed(1)
    x = torch.randn(n, 24)
    sage = fnn.SAGEConv(24, 48).cuda()
    out = sage(x.cuda(), ei.cuda()).detach().cpu()
    ref = gorc.sage_conv(x, ei, sage.lin_l.weight.detach().cpu(), sage.lin_l.bias.detach().cpu(), sage.lin_r.weight.detach().cpu())
    assert rel(out, ref) < 1e-4
    sage2 = fnn.SAGEConv(24, 8).cuda()  # out < in: transform-then-aggregate branch
    out = sage2(x.cuda(), ei.cuda()).detach().cpu()
    ref = gorc.sage_conv(x, ei, sage2.lin_l.weight.detach().cpu(), sage2.lin_l.bias.detach().cpu(), sage2.lin_r.weight.detach().cpu())
    assert rel(out, ref) < 1e-4
    mlp = torch.nn.Sequential(torch.nn.Linear(24, 32), torch.nn.ReLU())
    gin = fnn.GINConv(mlp, train_eps=True).cuda()
    with torch.no_grad():
        gin.eps.fill_(0.25)
    out = gin(x.cuda(), ei.cuda()).detach().cpu()
    ref = mlp.cpu()(gorc.gin_aggregate(x, ei, 0.25)).detach()
    assert rel(out, ref) < 1e-4
    ap = fnn.APPNP(K=10, alpha=0.1)
    out = ap(x.cuda(), ei.cuda()).cpu()
    assert rel(out, gorc.appnp(x, ei, 10, 0.1)) < 1e-4
    # class-wide signal -> the narrow kernel with the teleport term fused; wide signal -> the tiled kernel: same values,
    # and the propagation's gradient against the oracle's autograd
    for width in (7, 96):
        z = torch.randn(n, width)
        zg = z.cuda().requires_grad_(True)
        zc = z.clone().requires_grad_(True)
        o, r = ap(zg, ei.cuda()), gorc.appnp(zc, ei, 10, 0.1)
        assert rel(o.detach().cpu(), r.detach()) < 1e-4
        w = torch.randn(n, width)
        o.backward(w.cuda()); r.backward(w)
        assert rel(zg.grad.cpu(), zc.grad) < 1e-4, width


@pytest.mark.parametrize("dedup", [False, True])
def test_appnp_net_follows_the_sggc_model(mods, dedup):
    """network.APPNPNet (Baselines/SGGC/APPNP/networks.py:7-27) in eval mode vs the oracle: log-probabilities, loss, gradients;
    on a de-duplicated feature table (the MLP on the table, its output gathered to the union rows) it equals the materialised rows."""
    from fitgnn_amd import ops

    network, fnn, gorc = mods
    ei, n = graph(n=400, m=1500, seed=4)
    torch.manual_seed(2)
    n_table = 150
    table = torch.randn(n_table, 36)
    index = torch.randint(0, n_table, (n,))
    x = table[index]
    y = torch.randint(0, 7, (n,))
    tm = torch.rand(n) < 0.4
    args = argparse.Namespace(num_features=36, hidden=64, num_classes=7, K=10, alpha=0.1)
    model = network.APPNPNet(args).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    assert sorted(sd) == ["lin1.bias", "lin1.weight", "lin2.bias", "lin2.weight"]
    if dedup:
        out = model(table.cuda(), ei.cuda(), x_index=ops.RowIndex(index.cuda(), n_table))
    else:
        out = model(x.cuda(), ei.cuda())
    loss = torch.nn.functional.nll_loss(out[tm.cuda()], y.cuda()[tm.cuda()])
    loss.backward()
    o_ref, l_ref, g_ref = gorc.appnp_net_fwd_bwd(sd, x, ei, y, K=10, alpha=0.1, train_mask=tm)
    assert rel(out.detach().cpu(), o_ref) < 1e-4
    assert float(loss.detach()) == pytest.approx(float(l_ref), rel=1e-5)
    for k, p in model.named_parameters():
        assert rel(p.grad.cpu(), g_ref[k]) < 1e-3, k


@pytest.mark.parametrize("train", [False, True])
def test_classify_node_logits_loss_grads(mods, train):
    network, fnn, gorc = mods
    ei, n = graph(n=500, m=1500, seed=7)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=50, hidden=128, num_classes=6)
    torch.manual_seed(2)
    model = network.Classify_node(args).cuda()
    assert sorted(model.state_dict()) == ["conv.0.bias", "conv.0.lin.weight", "conv.1.bias", "conv.1.lin.weight",
                                          "lt1.bias", "lt1.weight"]
    x = torch.rand(n, 50)
    y = torch.randint(0, 6, (n,))
    tm = torch.rand(n) < 0.3
    masks = None
    if train:
        model.train()
        masks = [(torch.rand(n, 128) > 0.5).to(torch.uint8) for _ in range(2)]
        model._inject_masks = [m.cuda() for m in masks]
    else:
        model.eval()
    out = model(x.cuda(), ei.cuda())
    loss = torch.nn.functional.nll_loss(out[tm.cuda()], y.cuda()[tm.cuda()])
    loss.backward()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, x, ei, y, num_layers=2, train_mask=tm, masks=masks)
    assert rel(out.detach().cpu(), o_ref) < 1e-4
    assert abs(float(loss) - float(l_ref)) < 1e-4 * abs(float(l_ref))
    for k, p in model.named_parameters():
        assert rel(p.grad.cpu(), g_ref[k]) < 1e-3, k


@pytest.mark.parametrize("F_", [512, 36])
@pytest.mark.parametrize("with_rows", [False, True])
def test_segment_pools_equal_the_scatter_pools(mods, F_, with_rows):
    """global_mean_pool / global_max_pool on the segment kernels (fitgnn_segment_sum / _max / _expand_f32; sorted batch vector, the
    x[mask] gather of the *_gs models folded in) against the scatter forms: values and input gradients; empty graphs included."""
    network, fnn, gorc = mods
    torch.manual_seed(7)
    n, G = 700, 40
    sizes = torch.randint(0, 30, (G,))
    sizes[3] = 0
    sizes[-1] = 0
    batch_all = torch.repeat_interleave(torch.arange(G), sizes)
    n = int(batch_all.numel())
    x = torch.randn(n, F_)
    if with_rows:
        keep = torch.rand(n) < 0.6
        rows = torch.nonzero(keep).flatten()
        batch = batch_all[rows]
    else:
        rows, batch = None, batch_all
    w = torch.randn(G, F_)
    for pool, ref_pool in ((fnn.global_mean_pool, "mean"), (fnn.global_max_pool, "max")):
        xg = x.cuda().requires_grad_(True)
        out = pool(xg, batch.cuda(), G, rows=None if rows is None else rows.cuda())
        xr = x.clone().requires_grad_(True)
        sel = xr if rows is None else xr[rows]
        if ref_pool == "mean":
            ref = torch.zeros(G, F_).index_add_(0, batch, sel) / torch.bincount(batch, minlength=G).clamp(min=1).unsqueeze(1)
        else:
            ref = torch.full((G, F_), float("-inf")).scatter_reduce(0, batch.unsqueeze(1).expand_as(sel), sel, reduce="amax", include_self=True)
        finite = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(out.detach().cpu()), finite)
        assert rel(torch.where(finite, out.detach().cpu(), torch.zeros(())), torch.where(finite, ref.detach(), torch.zeros(()))) < 1e-6
        (out * torch.where(finite, w, torch.zeros(())).cuda()).nan_to_num(0.0, 0.0, 0.0).sum().backward()
        (torch.where(finite, ref, torch.zeros(())) * w).sum().backward()
        assert rel(xg.grad.cpu(), xr.grad) < 1e-6


def test_graph_level_models(mods):
    network, fnn, gorc = mods
    from types import SimpleNamespace

    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=32, num_classes=1)
    torch.manual_seed(3)
    model = network.Regress_graph_gs(args).cuda().eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(0)
    set_gs, bt, ref_rows = [], [], []
    for gi in range(5):
        gs = []
        for s in range(int(rng.integers(1, 4))):
            ei, n = graph(n=int(rng.integers(3, 9)), m=12, seed=gi * 10 + s)
            x = torch.rand(n, 11)
            mask = torch.rand(n) < 0.7
            mask[0] = True
            gs.append(SimpleNamespace(x=x, edge_index=ei, mask=mask))
            h = x
            for i in range(2):
                h = torch.nn.functional.elu(gorc.gcn_conv(h, ei, sd[f"conv.{i}.lin.weight"], sd[f"conv.{i}.bias"]))
            ref_rows.append(h[mask])
            bt += [gi] * int(mask.sum())
        set_gs.append(gs)
    bt = torch.tensor(bt)
    out = model(set_gs, bt.cuda()).detach().cpu()
    H = torch.cat(ref_rows)
    pooled = torch.stack([H[bt == gi].mean(0) for gi in range(5)])
    ref = pooled @ sd["lt1.weight"].t() + sd["lt1.bias"]
    assert rel(out, ref) < 1e-4


@pytest.mark.parametrize("C", [512, 96, 7])
def test_gatconv_forward_backward(mods, C):
    network, fnn, gorc = mods
    ei, n = graph(n=400, m=1600, seed=11)
    ei = torch.cat([ei, torch.tensor([[5, 9], [5, 9]])], 1)  # explicit self loops are replaced, not doubled
    torch.manual_seed(3)
    conv = fnn.GATConv(40, C).cuda()
    with torch.no_grad():
        conv.bias.normal_()
    assert sorted(conv.state_dict()) == ["att_dst", "att_src", "bias", "lin.weight"]
    x = torch.randn(n, 40)
    xg = x.cuda().requires_grad_(True)
    out = conv(xg, ei.cuda())
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in conv.state_dict().items()}
    xc = x.clone().requires_grad_(True)
    ref = gorc.gat_conv(xc, ei, P["lin.weight"], P["att_src"], P["att_dst"], P["bias"])
    assert rel(out.detach().cpu(), ref.detach()) < 1e-4
    gout = torch.randn(n, C)
    out.backward(gout.cuda())
    ref.backward(gout)
    assert rel(xg.grad.cpu(), xc.grad) < 1e-3
    for k, p in conv.named_parameters():
        assert rel(p.grad.cpu().reshape(-1), P[k].grad.reshape(-1)) < 1e-3, k


def test_classify_node_with_gat_layers(mods):
    network, fnn, gorc = mods
    ei, n = graph(n=300, m=900, seed=13)
    args = argparse.Namespace(num_layers1=2, layer_name="GATConv", num_features=20, hidden=64, num_classes=4)
    torch.manual_seed(4)
    model = network.Classify_node(args).cuda().eval()
    x = torch.rand(n, 20)
    out = model(x.cuda(), ei.cuda()).detach().cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    h = x
    for i in range(2):
        h = torch.nn.functional.elu(gorc.gat_conv(h, ei, sd[f"conv.{i}.lin.weight"], sd[f"conv.{i}.att_src"],
                                                  sd[f"conv.{i}.att_dst"], sd[f"conv.{i}.bias"]))
    ref = torch.log_softmax(h @ sd["lt1.weight"].t() + sd["lt1.bias"], dim=1)
    assert rel(out, ref) < 1e-4


@pytest.mark.parametrize("layer_name", ["GCNConv", "GATConv"])
def test_dedup_first_layer_equals_materialised_rows(mods, layer_name):
    """Layer 0 on the de-duplicated table + SpMM row indirection == layer 0 on the gathered union rows (GCN; GAT: the Linear and the
    score dots on the table, aggregation / SDDMM / softmax through the indirection, the copies' gradients summed per node)."""
    network, fnn, gorc = mods
    from fitgnn_amd import ops

    ei, n = graph(n=600, m=1800, seed=21)
    torch.manual_seed(5)
    N0 = 150
    idx = torch.randint(0, N0, (n,))
    idx[:N0] = torch.arange(N0)
    Xt = torch.rand(N0, 40)
    args = argparse.Namespace(num_layers1=2, layer_name=layer_name, num_features=40, hidden=128, num_classes=5)
    model = network.Classify_node(args).cuda().train()
    masks = [(torch.rand(n, 128) > 0.5).to(torch.uint8).cuda() for _ in range(2)]
    model._inject_masks = masks
    y = torch.randint(0, 5, (n,)).cuda()
    ridx = ops.RowIndex(idx.cuda(), N0)
    out_d = model(Xt.cuda(), ei.cuda(), x_index=ridx)
    torch.nn.functional.nll_loss(out_d, y).backward()
    g_d = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    out_m = model(Xt[idx].cuda(), ei.cuda())
    torch.nn.functional.nll_loss(out_m, y).backward()
    assert rel(out_d.detach().cpu(), out_m.detach().cpu()) < 1e-5
    for k, p in model.named_parameters():
        assert rel(g_d[k].cpu(), p.grad.cpu()) < 1e-4, k


def test_real_cora_trained_checkpoint(mods):
    """The reference's own trained Cora GCN (SGGC checkpoint, tests/golden/gcn_cora_sggc.npz) through the HIP GCNConv:
    log-probabilities within 1e-4 of the frozen oracle output, identical test accuracy."""
    import os

    import scipy.sparse as sp

    network, fnn, gorc = mods
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gcn_cora_sggc.npz"))
    X = sp.coo_matrix((d["x_val"], (d["x_row"], d["x_col"])), shape=tuple(d["x_shape"])).toarray()
    x = torch.nn.functional.normalize(torch.from_numpy(X), p=1).cuda()
    ei = torch.from_numpy(d["edge_index"].astype(np.int64)).cuda()
    c1, c2 = fnn.GCNConv(1433, 64).cuda(), fnn.GCNConv(64, 7).cuda()
    c1.load_state_dict({"lin.weight": torch.from_numpy(d["w:conv1.lin.weight"]), "bias": torch.from_numpy(d["w:conv1.bias"])})
    c2.load_state_dict({"lin.weight": torch.from_numpy(d["w:conv2.lin.weight"]), "bias": torch.from_numpy(d["w:conv2.bias"])})
    with torch.no_grad():
        logits = torch.log_softmax(c2(torch.relu(c1(x, ei)), ei), dim=1).cpu()
    ref = torch.from_numpy(d["logits"])
    assert float((logits - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    pred = logits.argmax(1).numpy()
    assert float((pred[d["test_idx"]] == d["y"][d["test_idx"]]).mean()) == pytest.approx(float(d["test_acc"]), abs=1e-3)


def _subgraph_batches(seed=0):
    """A small Gs: 40 clusters over a 400-node graph with extra nodes, as loader batches of 8 subgraphs."""
    from fitgnn_amd import data as fdata

    ei = fdata.synthetic_graph(400, 900, seed=seed)
    rng = np.random.default_rng(seed)
    assign = rng.integers(0, 40, size=400); assign[:40] = np.arange(40)
    sub = fdata.assemble_subgraphs(ei, 400, assign, 40, extra_node=True)
    X = torch.from_numpy(rng.standard_normal((400, 24)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 4, size=400))
    tm = torch.from_numpy(rng.random(400) < 0.3)
    batch = fdata.SubgraphBatch(sub, X, y, tm, device="cuda")
    cpu = []
    for r0, r1 in batch.slice_batches(8):
        e = batch.edge_index.cpu()
        k = (e[0] >= r0) & (e[0] < r1)
        cpu.append(dict(x=batch.x[r0:r1].cpu(), edge_index=e[:, k] - r0, y=batch.y[r0:r1].cpu(), train_mask=batch.train_mask[r0:r1].cpu()))
    return batch, cpu


@pytest.mark.parametrize("method", ["GD", "MB", "MB-captured"])
def test_trainers_follow_the_reference_step_functions(mods, method):
    """GDTrainer / MBTrainer vs the oracle's restatement of run.py:177-215 / :217-252 over three epochs (dropout off):
    same reported loss and the same weights (MB: including the gradient accumulation across batches)."""
    from fitgnn_amd import train

    network, fnn, gorc = mods
    batch, cpu = _subgraph_batches()
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=32, num_classes=4)
    torch.manual_seed(5)
    model = network.Classify_node(args).cuda()
    model.dropout_p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if method == "GD":
        tr = train.GDTrainer(model, batch, lr=0.01, weight_decay=5e-4)
    else:
        tr = train.MBTrainer(model, batch, batch_size=8, lr=0.01, weight_decay=5e-4, capture=method.endswith("captured"))
    state = None
    for epoch in range(3):
        got = float(tr.step())
        if method == "GD":
            want, sd, state = gorc.gd_train_step(sd, cpu, adam_state=state, masks=None)
        else:
            want, sd, state = gorc.mb_train_epoch(sd, cpu, adam_state=state)
        assert got == pytest.approx(float(want), rel=2e-4), (epoch, got, float(want))
    for k, v in model.state_dict().items():
        assert rel(v.detach().cpu(), sd[k]) < 2e-3, k


@pytest.mark.parametrize("kind,extra", [("gs", True), ("gs", False), ("gc", False)])
def test_graph_level_training_follows_the_reference_loops(mods, kind, extra):
    """GraphSet + GraphTrainer (one block-diagonal pass per batch) vs the oracle's literal restatement: per-subgraph
    conv stacks with the reference's first-k row mask (network.py:189-204) and the graph_train_Gs / _Gc step loops
    (run.py:254-304: one zero_grad per epoch, targets through .long())."""
    from fitgnn_amd import graph_data, train

    network, fnn, gorc = mods
    mol = graph_data.synthetic_molecules(48, seed=3)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=extra, device="cuda")
    assert int(gset.gs_mask.sum()) == int(mol["node_ptr"][-1])   # sum of k over subgraphs = number of nodes
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=32, num_classes=1)
    torch.manual_seed(11)
    model = (network.Regress_graph_gs if kind == "gs" else network.Regress_graph_gc)(args).cuda()
    model.dropout_p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = train.GraphTrainer(model, gset, list(range(48)), kind=kind, batch_size=16, prop=2)
    # the same three batches for the oracle, in the reference's data layout
    cpu_batches = []
    y = torch.from_numpy(mol["y"])
    for g0 in range(0, 48, 16):
        if kind == "gs":
            set_gs = []
            for g in range(g0, g0 + 16):
                subs = []
                for c in range(int(gset.cluster_ptr[g]), int(gset.cluster_ptr[g + 1])):
                    r0, r1 = int(gset.sub_ptr[c]), int(gset.sub_ptr[c + 1])
                    e = gset.gs_edge_index.cpu()
                    k = (e[0] >= r0) & (e[0] < r1)
                    subs.append(dict(x=gset.gs_x[r0:r1].cpu(), edge_index=e[:, k] - r0, mask=gset.gs_mask[r0:r1].cpu()))
                set_gs.append(subs)
            bt = torch.repeat_interleave(torch.arange(16), torch.from_numpy(np.diff(mol["node_ptr"][g0:g0 + 17])))
            cpu_batches.append((set_gs, bt, y[g0:g0 + 16]))
        else:
            r0, r1 = int(gset.cluster_ptr[g0]), int(gset.cluster_ptr[g0 + 16])
            e = gset.gc_edge_index.cpu()
            k = (e[0] >= r0) & (e[0] < r1)
            cpu_batches.append((gset.gc_x[r0:r1].cpu(), e[:, k] - r0, gset.gc_graph[r0:r1].cpu() - g0, 16, y[g0:g0 + 16]))
    fwd = gorc.regress_graph_gs_forward if kind == "gs" else gorc.regress_graph_gc_forward
    state = None
    for epoch in range(2):
        got = float(tr.step())
        want, sd, state = gorc.graph_train_epoch(sd, cpu_batches, fwd, prop=2, adam_state=state)
        assert got == pytest.approx(want, rel=5e-4), (epoch, got, want)
    for k, v in model.state_dict().items():
        assert rel(v.detach().cpu(), sd[k]) < 3e-3, k


@pytest.mark.parametrize("kind", ["gs", "gc"])
def test_captured_graph_steps_equal_eager_steps(mods, kind):
    """GraphTrainer(capture=True): every batch step replayed from a hipGraph == the eager steps (dropout off: same
    losses and weights); with dropout on, successive replays draw different patterns (device-resident seeds)."""
    from fitgnn_amd import graph_data, train

    network, fnn, gorc = mods
    mol = graph_data.synthetic_molecules(64, seed=5)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=32, num_classes=1)
    cls = network.Regress_graph_gs if kind == "gs" else network.Regress_graph_gc
    torch.manual_seed(3)
    m1 = cls(args).cuda()
    m2 = cls(args).cuda()
    m2.load_state_dict(m1.state_dict())
    m1.dropout_p = m2.dropout_p = 0.0
    t1 = train.GraphTrainer(m1, gset, list(range(64)), kind=kind, batch_size=16, prop=1)
    t2 = train.GraphTrainer(m2, gset, list(range(64)), kind=kind, batch_size=16, prop=1, capture=True)
    for epoch in range(3):
        a, b = float(t1.step()), float(t2.step())
        assert a == pytest.approx(b, rel=1e-5), (epoch, a, b)
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert rel(w, v) < 1e-4, k
    # dropout on: two replays of the same captured step must not repeat the pattern
    m2.dropout_p = 0.5
    t3 = train.GraphTrainer(m2, gset, list(range(16)), kind=kind, batch_size=16, prop=1, capture=True, lr=0.0)
    l1, l2 = float(t3.step()), float(t3.step())
    assert l1 != l2


def test_flat_artefact_round_trip(mods, tmp_path):
    """store.save_gs / load_gs: the subgraph union written as flat .npy arrays and mapped back gives the same batch
    (rows, edges, features, labels, masks) and the same model output."""
    from fitgnn_amd import store

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=4)
    store.save_gs(str(tmp_path / "a"), batch, extra_meta={"dataset": "t"})
    got, meta = store.load_gs(str(tmp_path / "a"))
    assert meta["format"] == store.FORMAT and meta["n_rows"] == batch.n_rows and meta["dataset"] == "t"
    assert np.array_equal(got.ptr, batch.ptr)
    for k in ("node_id", "core", "y", "train_mask", "train_idx", "x"):
        assert torch.equal(getattr(got, k), getattr(batch, k)), k
    ea = set(map(tuple, batch.edge_index.t().tolist())); eb = set(map(tuple, got.edge_index.t().tolist()))
    assert ea == eb
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=32, num_classes=4)
    torch.manual_seed(0)
    model = network.Classify_node(args).cuda().eval()
    with torch.no_grad():
        assert torch.allclose(model(batch.x, batch.edge_index), model(got.x, got.edge_index), rtol=1e-5, atol=1e-6)


def test_flat_adam_equals_torch_adam(mods):
    """train.FlatAdam (one kernel over the flat parameter / gradient / moment buffers) == torch.optim.Adam with the
    reference's settings (run.py:344), step by step, and speaks its state_dict format."""
    from fitgnn_amd import train

    torch.manual_seed(0)
    shapes = [(64, 24), (64,), (64, 64), (64,), (5, 64), (5,)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref = torch.optim.Adam(qs, lr=0.01, weight_decay=5e-4)
    flat = train.FlatGrads(ps)
    opt = train.FlatAdam(flat, lr=0.01, weight_decay=5e-4)
    for it in range(5):
        for p, q in zip(ps, qs):
            g = torch.randn_like(q)
            p.grad.copy_(g)
            q.grad = g.clone()
        opt.step(); ref.step()
        for p, q in zip(ps, qs):
            assert rel(p.detach(), q.detach()) < 1e-6, it
    sd = opt.state_dict()
    assert set(sd["state"]) == set(range(6)) and float(sd["state"][0]["step"]) == 5.0
    ref2 = torch.optim.Adam(qs, lr=0.01, weight_decay=5e-4)
    ref2.load_state_dict(sd)          # torch accepts it
    opt.load_state_dict(ref.state_dict())
    assert float(opt.step_count[0]) == 5.0 and rel(opt._views(opt.m)[2], ref.state[qs[2]]["exp_avg"]) < 1e-6


def test_pruned_last_layer_equals_full_evaluation(mods):
    """GDTrainer(prune_unused_rows=True) -- last layer as (A_hat[loss rows] h) W^T: rows nobody reads are not aggregated -- gives
    the same loss and, after three epochs, the same weights as the full evaluation (dropout off); RowSubset's two
    patterns are adjoint."""
    from fitgnn_amd import ops, train
    from fitgnn_amd.csr import RowSubset

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=6)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=32, num_classes=4)
    torch.manual_seed(8)
    m1 = network.Classify_node(args).cuda(); m1.dropout_p = 0.0
    m2 = network.Classify_node(args).cuda(); m2.dropout_p = 0.0
    m2.load_state_dict(m1.state_dict())
    t1 = train.GDTrainer(m1, batch, lr=0.01, weight_decay=5e-4)
    t2 = train.GDTrainer(m2, batch, lr=0.01, weight_decay=5e-4, prune_unused_rows=True)
    # the pruned step = the aggregate-first last layer with its forward aggregation on the loss rows alone (A_hat[train rows, :] h)
    assert t2.prune_forward and t2.sub is not None and t2.sub.m == int(batch.train_idx.numel())
    for epoch in range(3):
        a, b = float(t1.step()), float(t2.step())
        assert a == pytest.approx(b, rel=2e-5), (epoch, a, b)
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert rel(w, v) < 2e-4, k
    sub = t2.sub
    X, Z = torch.randn(batch.n_rows, 32, device="cuda"), torch.randn(sub.m, 32, device="cuda")
    Y = ops.SpMMRows.apply(X, sub, ops.DEFAULT)
    full = ops.spmm_graph(batch.graph, X).index_select(0, sub.rows)
    assert rel(Y, full) < 1e-6
    XT = ops.spmm_raw(sub.t.rowptr, sub.t.col, sub.t.val, sub.t.tiles, Z, sub.n, window_rows=sub.window_rows)
    lhs, rhs = float((Y.double() * Z.double()).sum()), float((X.double() * XT.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * (abs(lhs) + abs(rhs) + 1)


@pytest.mark.parametrize("H,C,n_rows", [(512, 3, 1000), (512, 47, 5001), (64, 7, 13), (128, 60, 257), (512, 5, 1)])
def test_head_on_selected_rows(mods, H, C, n_rows):
    """fitgnn_head_rows_f32: the output head on the loss rows only equals out[rows] @ Wl^T + bl (fp32 tolerance: a different
    summation order than the library's GEMM), leaves every other row of y zero, handles a ragged last group and a row stride."""
    from fitgnn_amd import ops

    torch.manual_seed(H + C)
    R = 4 * n_rows + 3
    base = torch.randn(R, H + 4, device="cuda")
    out = base[:, :H]                                  # row stride H + 4
    Wl, bl = torch.randn(C, H, device="cuda") / H ** 0.5, torch.randn(C, device="cuda")
    rows = torch.randperm(R, device="cuda")[:n_rows].sort().values
    assert ops.head_rows_supported(out, Wl)
    y = ops.head_rows(out, rows, Wl, bl)
    ref = (out.double() @ Wl.double().t() + bl.double()).float()
    assert rel(y.index_select(0, rows), ref.index_select(0, rows)) < 2e-6
    rest = torch.ones(R, dtype=torch.bool, device="cuda"); rest[rows] = False
    assert float(y[rest].abs().max()) == 0.0
    y0 = ops.head_rows(out, rows, Wl, None)
    assert rel(y0.index_select(0, rows), (ref - bl).index_select(0, rows)) < 2e-6


@pytest.mark.parametrize("layers,dedup", [(2, False), (3, False), (2, True)])
def test_two_hop_backward_in_the_model(mods, layers, dedup):
    """embed_and_head(loss_rows=..., compact_logits=True) on a batch that runs on the whole-subgraph kernel: with
    OpConfig.two_hop_backward the layer below the last one (a plain fused GCN layer, or layer 0 on a de-duplicated table) receives
    A_hat^T dZ from the last layer's backward and skips its own SpMM -- same logits, same loss, the same weight gradients BIT FOR BIT as
    with the two separate launches (bias gradient of that layer: fp32 summation order), dropout on (the same injected masks)."""
    from fitgnn_amd import csr, ops

    network, fnn, gorc = mods
    rng = np.random.default_rng(4)
    sizes = [60, 130, 18, 40, 300, 25, 90]
    src, dst, off = [], [], 0
    for s_ in sizes:   # stars with two centres and a few leaf -- leaf edges
        for h in range(2):
            leaves = np.arange(2, s_)
            src += [off + h] * len(leaves) + (off + leaves).tolist()
            dst += (off + leaves).tolist() + [off + h] * len(leaves)
        a, b = rng.integers(2, s_, size=s_ // 4), rng.integers(2, s_, size=s_ // 4)
        k = a != b
        src += (off + a[k]).tolist() + (off + b[k]).tolist()
        dst += (off + b[k]).tolist() + (off + a[k]).tolist()
        off += s_
    n = off
    ei = torch.tensor(np.unique(np.array([src, dst]), axis=1), dtype=torch.long).cuda()
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    g = csr.CSRGraph(ei, n, mode="gcn", ptr=ptr, block_limit=4096)
    assert g.t.blocks is not None
    csr.register(ei, g)
    rows = torch.cat([torch.arange(2) + o for o in ptr[:-1]]).cuda()
    args = argparse.Namespace(num_layers1=layers, layer_name="GCNConv", num_features=24, hidden=64, num_classes=5)
    torch.manual_seed(3)
    m = network.Classify_node(args).cuda()
    m.train()
    m.dropout_p = 0.5
    y = torch.randint(0, 5, (n,)).cuda()
    if dedup:
        n_table = n // 3
        index = torch.randint(0, n_table, (n,)).cuda()
        x, x_index = torch.randn(n_table, 24).cuda(), ops.RowIndex(index, n_table)
    else:
        x, x_index = torch.randn(n, 24).cuda(), None
    res = []
    for two in (True, False):
        cfg = ops.OpConfig(two_hop_backward=two, profile=[])
        m.set_op_config(cfg)
        m.zero_grad()
        torch.manual_seed(11)
        m._inject_masks = [(torch.rand(n, 64, device="cuda") > 0.5).to(torch.uint8) for _ in range(layers)]
        z = m.embed_and_head(x, ei, x_index, loss_rows=rows, compact_logits=True)
        assert z.shape == (rows.numel(), 5)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(z, 1), y.index_select(0, rows), reduction="sum")
        loss.backward()
        kinds = [k for _, _, k in cfg.profile]
        assert ("two_hop" in kinds) == two, kinds
        res.append((z.detach().clone(), float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters()}))
    m._inject_masks = None
    m.set_op_config(ops.DEFAULT)
    (z1, l1, g1), (z0, l0, g0) = res
    assert torch.equal(z1, z0) and l1 == l0
    below = "conv.%d.bias" % (layers - 2)
    for k in g0:
        if k == below:
            assert rel(g1[k], g0[k]) < 1e-5, k
        else:
            assert torch.equal(g1[k], g0[k]), k


def test_loss_rows_hint_changes_nothing_the_loss_sees(mods):
    """embed_and_head(loss_rows=...) in its three forms -- last layer aggregate-first with the dense part on the loss rows
    (default), transform-first with the head and a compact dZ on the loss rows, transform-first with only the head's weight
    gradients restricted -- gives the logits on those rows, the loss and every gradient of the plain full evaluation; with
    dropout on (injected masks: every form drops the same entries) as well."""
    from fitgnn_amd import ops

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=9)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=64, num_classes=5)
    torch.manual_seed(3)
    m = network.Classify_node(args).cuda()
    m.train()
    idx = batch.train_idx
    torch.manual_seed(5)
    masks = [(torch.rand(batch.n_rows, 64, device="cuda") > 0.5).to(torch.uint8) for _ in range(2)]
    forms = [("full", None, ops.OpConfig()),
             ("aggregate-first on loss rows", idx, ops.OpConfig()),
             ("compact dZ", idx, ops.OpConfig(last_layer_on_loss_rows=False)),
             ("head gradients only", idx, ops.OpConfig(last_layer_on_loss_rows=False, compact_head_backward=False))]
    for p_drop in (0.0, 0.5):
        m.dropout_p = p_drop
        m._inject_masks = masks if p_drop > 0 else None
        res = []
        for name, hint, cfg in forms:
            m.set_op_config(cfg)
            m.zero_grad()
            z = m.embed_and_head(batch.x, batch.edge_index, loss_rows=hint)
            loss = torch.nn.functional.nll_loss(torch.log_softmax(z.index_select(0, idx), 1), batch.y.index_select(0, idx), reduction="sum")
            loss.backward()
            res.append((name, z.detach().index_select(0, idx), float(loss), [p.grad.clone() for p in m.parameters()]))
        _, z0, l0, g0 = res[0]
        for name, z1, l1, g1 in res[1:]:
            assert rel(z1, z0) < 1e-4, (name, p_drop)
            assert l1 == pytest.approx(l0, rel=1e-5), (name, p_drop)
            for a, b in zip(g1, g0):
                assert rel(a, b) < 2e-4, (name, p_drop)
    m._inject_masks = None
    m.set_op_config(ops.DEFAULT)
    # compact_logits: the same logits as a [len(rows), C] matrix in the order of loss_rows, and the same gradients through it
    m.dropout_p = 0.0
    m.zero_grad()
    zc = m.embed_and_head(batch.x, batch.edge_index, loss_rows=idx, compact_logits=True)
    assert zc.shape == (idx.numel(), 5)
    torch.nn.functional.nll_loss(torch.log_softmax(zc, 1), batch.y.index_select(0, idx), reduction="sum").backward()
    gc = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    zf = m.embed_and_head(batch.x, batch.edge_index, loss_rows=idx)
    torch.nn.functional.nll_loss(torch.log_softmax(zf.index_select(0, idx), 1), batch.y.index_select(0, idx), reduction="sum").backward()
    assert torch.equal(zc, zf.index_select(0, idx))
    for a, b in zip(gc, [p.grad for p in m.parameters()]):
        assert rel(a, b) < 1e-6


@pytest.mark.parametrize("dedup", [False, True])
@pytest.mark.parametrize("classes", [5, 47])
def test_gat_last_layer_on_the_loss_rows_changes_nothing_the_loss_sees(mods, dedup, classes):
    """Classify_node with GATConv layers: embed_and_head(loss_rows=...) evaluates the last attention layer aggregate-first with its
    dense part on the loss rows (ops.FusedGATLastLayerRows: scores from x . (W^T att), sum_j alpha_ij x_j, then W / bias / ELU /
    dropout / head on the kept rows; backward edge passes on those rows' entries only) -- the same logits on those rows, loss and
    gradients of every parameter as the plain full evaluation, dropout off and on (injected masks), compact logits included."""
    from fitgnn_amd import ops

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=12)
    hidden = 64 if classes <= 16 else 256   # (the fused head backward takes a head as wide as the last 256-column slab has lanes)
    args = argparse.Namespace(num_layers1=2, layer_name="GATConv", num_features=24, hidden=hidden, num_classes=classes)
    torch.manual_seed(6)
    m = network.Classify_node(args).cuda().train()
    idx = batch.train_idx
    y = (batch.y % classes).index_select(0, idx)
    torch.manual_seed(7)
    masks = [(torch.rand(batch.n_rows, hidden, device="cuda") > 0.5).to(torch.uint8) for _ in range(2)]

    def run(cfg, hint, compact=False):
        m.set_op_config(cfg)
        m.zero_grad()
        if dedup:
            z = m.embed_and_head(batch.x_table, batch.edge_index, batch.row_index, loss_rows=hint, compact_logits=compact)
        else:
            z = m.embed_and_head(batch.x, batch.edge_index, loss_rows=hint, compact_logits=compact)
        if hint is not None and cfg.last_layer_on_loss_rows:   # the aggregate-first node ran, not a silent fall-back
            assert "FusedGATLastLayerRows" in type(z.grad_fn).__name__, type(z.grad_fn).__name__
        zl = z if (compact and z.shape[0] == idx.numel()) else z.index_select(0, idx)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(zl, 1), y, reduction="sum")
        loss.backward()
        return zl.detach(), float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}

    for p_drop in (0.0, 0.5):
        m.dropout_p = p_drop
        m._inject_masks = masks if p_drop > 0 else None
        z0, l0, g0 = run(ops.OpConfig(last_layer_on_loss_rows=False), None)
        for compact in (False, True):
            z1, l1, g1 = run(ops.OpConfig(), idx, compact)
            assert rel(z1, z0) < 1e-4, (p_drop, compact)
            assert l1 == pytest.approx(l0, rel=1e-5), (p_drop, compact)
            for k in g0:
                assert rel(g1[k], g0[k]) < 5e-4, (k, p_drop, compact)
    m._inject_masks = None
    m.set_op_config(ops.DEFAULT)


def test_forward_epilogue_on_compact_rows_is_the_spmm_epilogue(mods):
    """fitgnn_epilogue_fwd_rows_f32 on gathered rows == the SpMM kernel's store epilogue on the same rows, bit for bit, with the
    seed-hashed dropout pattern of the ORIGINAL rows."""
    from fitgnn_amd import _lib, ops
    from fitgnn_amd.csr import CSRGraph

    ei, n = graph(500, 2000, seed=4)
    g = CSRGraph(ei.cuda(), n, mode="gcn")
    torch.manual_seed(1)
    X, b = torch.randn(n, 64, device="cuda"), torch.randn(64, device="cuda")
    flags = _lib.EPI_BIAS | _lib.EPI_ELU | _lib.EPI_DROPOUT
    full = ops.spmm_graph(g, X, bias=b, epilogue=flags, p=0.4, seed=1234)
    plain = ops.spmm_graph(g, X)
    rows = torch.randperm(n, device="cuda")[:177].sort().values
    zc = plain.index_select(0, rows).contiguous()
    ops.epilogue_fwd_rows_(zc, rows, b, flags, p=0.4, seed=1234)
    assert torch.equal(zc, full.index_select(0, rows))


def test_graph_trainer_reshuffle_option(mods):
    """GraphTrainer(reshuffle=True) re-draws the graph order every epoch (run.py:710 shuffle=True): batches change between
    epochs, every graph is still visited exactly once per epoch, and the loss keeps decreasing."""
    from fitgnn_amd import graph_data, train

    network, fnn, gorc = mods
    mol = graph_data.synthetic_molecules(96, seed=9)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=32, num_classes=1)
    torch.manual_seed(1)
    model = network.Regress_graph_gs(args).cuda()
    tr = train.GraphTrainer(model, gset, list(range(96)), kind="gs", batch_size=32, prop=0, lr=0.005, reshuffle=True)
    seen, losses = [], []
    for epoch in range(6):
        losses.append(float(tr.step()))
        seen.append(torch.cat([b["y"][:, 0] for b in tr.batches]).cpu())
    assert not torch.equal(seen[0], seen[1])                                    # a different order ...
    assert torch.equal(seen[0].sort().values, seen[1].sort().values)            # ... of the same graphs
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("layers,dedup", [(2, False), (3, False), (2, True)])
@pytest.mark.parametrize("inject", [True, False])
def test_dx_gemm_with_the_previous_layers_epilogue(mods, layers, dedup, inject):
    """Consecutive fused GCN layers: the backward GEMM dH @ W applies the previous layer's ELU'/dropout' in its epilogue
    (ops.EpilogueLink, csrc/gemm_nt.hip EPI).  Same gradients as the two-kernel path and as the oracle."""
    network, fnn, gorc = mods
    from fitgnn_amd import ops

    n = 1500   # the GEMM kernels take over from 1024 rows
    ei, n = graph(n=n, m=4500, seed=31)
    args = argparse.Namespace(num_layers1=layers, layer_name="GCNConv", num_features=64, hidden=128, num_classes=5)
    torch.manual_seed(9)
    model = network.Classify_node(args).cuda().train()
    x = torch.rand(n, 64)
    y = torch.randint(0, 5, (n,))
    masks = [(torch.rand(n, 128) > 0.5).to(torch.uint8) for _ in range(layers)]
    ridx = None
    xin = x.cuda()
    if dedup:
        N0 = 400
        idx = torch.randint(0, N0, (n,)); idx[:N0] = torch.arange(N0)
        xt = torch.rand(N0, 64)
        x, xin, ridx = xt[idx], xt.cuda(), ops.RowIndex(idx.cuda(), N0)
    calls = []
    real = ops.gemm_nt_epilogue_bwd

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)

    grads = {}
    for fuse in (True, False):
        ops.gemm_nt_epilogue_bwd = counting
        model.set_op_config(ops.OpConfig(gemm_precision="high", fuse_dx_epilogue=fuse))   # per-model switch: no process-wide state
        try:
            model.zero_grad()
            if inject:
                model._inject_masks = [m.cuda() for m in masks]
            else:
                model._inject_masks = None
                torch.manual_seed(777)   # ops.next_seed draws from torch's generator: same dropout stream in both runs
            out = model(xin, ei.cuda(), x_index=ridx) if dedup else model(xin, ei.cuda())
            torch.nn.functional.nll_loss(out, y.cuda()).backward()
            grads[fuse] = {k: p.grad.clone() for k, p in model.named_parameters()}
        finally:
            ops.gemm_nt_epilogue_bwd = real
            model.set_op_config(ops.DEFAULT)
    assert len(calls) == layers - 1, "one fused GEMM per linked pair of layers, none with the switch off"
    for k in grads[True]:
        assert rel(grads[True][k].cpu(), grads[False][k].cpu()) < 1e-6, k
    if inject:
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        _, _, g_ref = gorc.classify_node_fwd_bwd(sd, x, ei, y, num_layers=layers, train_mask=torch.ones(n, dtype=torch.bool), masks=masks)
        for k, p in model.named_parameters():
            assert rel(grads[True][k].cpu(), g_ref[k]) < 1e-3, k


@pytest.mark.parametrize("hidden,classes,layers", [(256, 16, 3), (64, 2, 2), (32, 4, 2), (128, 47, 2), (256, 47, 2), (512, 48, 2), (256, 49, 2), (16, 7, 2), (32, 16, 2),
                                                   (320, 17, 2)])
def test_other_widths_through_the_gemm_kernels(mods, hidden, classes, layers):
    """Hidden sizes on either side of the GEMM kernels' limits (64 columns, K % 32) and head widths on either side of the fused
    head-backward kernel's (16 classes with its own weight gradient, 48 without: ogbn-products has 47) against the oracle,
    training mode with injected dropout masks, 4000 rows (a 16 x 1 grid of 256-row tiles: the 128 x 128 tile variant)."""
    network, fnn, gorc = mods
    ei, n = graph(n=4000, m=12000, seed=41)
    args = argparse.Namespace(num_layers1=layers, layer_name="GCNConv", num_features=96, hidden=hidden, num_classes=classes)
    torch.manual_seed(3)
    model = network.Classify_node(args).cuda().train()
    x = torch.rand(n, 96)
    y = torch.randint(0, classes, (n,))
    tm = torch.rand(n) < 0.4
    masks = [(torch.rand(n, hidden) > 0.5).to(torch.uint8) for _ in range(layers)]
    model._inject_masks = [m.cuda() for m in masks]
    out = model(x.cuda(), ei.cuda())
    loss = torch.nn.functional.nll_loss(out[tm.cuda()], y.cuda()[tm.cuda()])
    loss.backward()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, x, ei, y, num_layers=layers, train_mask=tm, masks=masks)
    assert rel(out.detach().cpu(), o_ref) < 1e-4
    assert abs(float(loss.detach()) - float(l_ref)) < 1e-4 * abs(float(l_ref))
    for k, p in model.named_parameters():
        assert rel(p.grad.cpu(), g_ref[k]) < 1e-3, k


@pytest.mark.parametrize("kind", ["gs", "gc", "orig"])
def test_batch_of_arbitrary_graph_ids_equals_per_graph_pieces(mods, kind):
    """GraphSet.batch_ids (two gathers from the per-graph pointers) == the concatenation of single-graph batch() pieces."""
    import types
    from fitgnn_amd import graph_data, train

    mol = graph_data.synthetic_molecules(40, seed=3)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    ids = [17, 3, 39, 0, 22, 23, 8]
    a = train._cat_pieces([gset.batch_ids(ids, kind)], kind, types)
    b = train._cat_pieces([gset.batch(g, g + 1, kind) for g in ids], kind, types)
    assert torch.equal(a["x"], b["x"]) and torch.equal(a["y"], b["y"])
    ea = a["edge_index"][:, torch.argsort(a["edge_index"][0] * 10 ** 6 + a["edge_index"][1])]
    eb = b["edge_index"][:, torch.argsort(b["edge_index"][0] * 10 ** 6 + b["edge_index"][1])]
    assert torch.equal(ea, eb)
    if kind == "gs":
        assert torch.equal(a["mask"], b["mask"]) and torch.equal(a["graph_of_masked"], b["graph_of_masked"])
    else:
        assert torch.equal(a["gc"].batch, b["gc"].batch)


def test_wide_unaligned_feature_table_runs_on_the_gemm_kernels(mods):
    """Real feature widths are not multiples of 32 (500, 1 433, 8 415).  For a wide, static table layer 0's two products run
    on a zero-padded copy (ops.padded_table) through the pre-split GEMM path: same outputs and gradients as the library path."""
    network, fnn, gorc = mods
    from fitgnn_amd import ops

    N0, F, H, n = 17000, 1100, 512, 20000
    ei, n = graph(n=n, m=60000, seed=51)
    torch.manual_seed(8)
    idx = torch.randint(0, N0, (n,)); idx[:N0] = torch.arange(N0)
    Xt = torch.rand(N0, F).cuda()
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=H, num_classes=5)
    model = network.Classify_node(args).cuda().train()
    masks = [(torch.rand(n, H) > 0.5).to(torch.uint8).cuda() for _ in range(2)]
    model._inject_masks = masks
    y = torch.randint(0, 5, (n,)).cuda()
    ridx = ops.RowIndex(idx.cuda(), N0)
    res = {}
    for wide in (True, False):
        model.set_op_config(ops.OpConfig(gemm_precision="high", pad_table_min_k=0 if wide else 10 ** 9))
        model.zero_grad()
        out = model(Xt, ei.cuda(), x_index=ridx)
        torch.nn.functional.nll_loss(out, y).backward()
        res[wide] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert getattr(Xt, "_fitgnn_pad")[1].shape == (N0, 1120)
    assert rel(res[True][0].cpu(), res[False][0].cpu()) < 1e-5
    for k in res[True][1]:
        assert res[True][1][k].shape == res[False][1][k].shape
        assert rel(res[True][1][k].cpu(), res[False][1][k].cpu()) < 1e-4, k


def test_feature_width_that_is_not_a_multiple_of_four_under_the_exact_policy(mods):
    """Cora's 1 433 features (8 415 at Physics): rows of the table and of W0 are not 16-byte aligned, so under the default exact-fp32
    policy layer 0's two products run against the zero-padded table and a zero-padded copy of W0 (ops.FusedGCNLayerDedup):
    outputs and gradients equal the fp32 library products' (both are fp32 GEMMs: agreement to accumulation-order rounding)."""
    network, fnn, gorc = mods
    from fitgnn_amd import ops

    N0, F, H, n = 3000, 1433, 128, 5000
    ei, n = graph(n=n, m=15000, seed=52)
    torch.manual_seed(9)
    idx = torch.randint(0, N0, (n,)); idx[:N0] = torch.arange(N0)
    Xt = torch.rand(N0, F).cuda()
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=H, num_classes=7)
    model = network.Classify_node(args).cuda().train()
    model._inject_masks = [(torch.rand(n, H) > 0.5).to(torch.uint8).cuda() for _ in range(2)]
    y = torch.randint(0, 7, (n,)).cuda()
    ridx = ops.RowIndex(idx.cuda(), N0)
    res = {}
    for prec in ("exact", "highest"):
        log = []
        model.set_op_config(ops.OpConfig(gemm_precision=prec, profile_gemm=log))
        model.zero_grad()
        out = model(Xt, ei.cuda(), x_index=ridx)
        torch.nn.functional.nll_loss(out, y).backward()
        res[prec] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, [e[2] for e in log])
    model.set_op_config(ops.DEFAULT)
    assert "gemm_f32_kernel[nt]" in res["exact"][2] and "gemm_f32_kernel[tn]" in res["exact"][2] and not res["highest"][2]
    assert getattr(Xt, "_fitgnn_pad")[1].shape == (N0, 1440)
    assert rel(res["exact"][0].cpu(), res["highest"][0].cpu()) < 2e-6
    for k in res["exact"][1]:
        assert res["exact"][1][k].shape == res["highest"][1][k].shape
        assert rel(res["exact"][1][k].cpu(), res["highest"][1][k].cpu()) < 2e-5, k


def test_row_index_caches_follow_the_index_tensor_not_its_address(mods):
    """embed_and_head(loss_rows=...) caches the compact-operand positions (and the per-entry table rows) on the graph.  The cache
    entry holds the index tensor it was built from and its version: a caller that rewrites its loss_rows IN PLACE (same
    address, same length) gets positions for the new rows, not the stale ones -- gradients equal the un-hinted evaluation's."""
    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=9)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=64, num_classes=5)
    torch.manual_seed(3)
    m = network.Classify_node(args).cuda()
    m.eval()
    core = torch.nonzero(batch.core).flatten()
    k = int(core.numel()) // 2
    first, second = core[:k].clone(), core[-k:].clone()
    assert not torch.equal(first, second)

    def grads(rows, hint):
        m.zero_grad()
        z = m.embed_and_head(batch.x, batch.edge_index, loss_rows=rows if hint else None, compact_logits=hint)
        zs = z if (hint and z.shape[0] == rows.numel()) else z.index_select(0, rows)
        torch.nn.functional.nll_loss(torch.log_softmax(zs, 1), batch.y.index_select(0, rows), reduction="sum").backward()
        return [p.grad.clone() for p in m.parameters()]

    rows = first.clone()
    g_first = grads(rows, True)
    rows.copy_(second)                       # same tensor, same address, new selection
    g_second = grads(rows, True)
    for a, b in zip(g_second, grads(second, False)):
        assert rel(a, b) < 2e-4
    for a, b in zip(g_first, grads(first, False)):
        assert rel(a, b) < 2e-4


@pytest.mark.parametrize("K,H,n,layers", [(11, 512, 1237, 2), (3, 64, 77, 2), (32, 128, 333, 1), (1, 16, 5, 2), (17, 256, 2050, 3)])
@pytest.mark.parametrize("mode", ["eval", "masks", "hashed"])
def test_narrow_first_layer_runs_aggregate_first_with_the_same_values(mods, K, H, n, layers, mode):
    """ops.FusedGCNLayerAggregatedInput (A_hat x formed once, fitgnn_dense_narrow_k_f32 / fitgnn_narrow_atb_f32) against the
    transform-first layer (OpConfig(narrow_input_first=False)) and against the oracle: logits, loss, every gradient."""
    network, fnn, gorc = mods
    from fitgnn_amd import ops

    ei, _ = graph(n=n, m=3 * n, seed=K + n)
    args = argparse.Namespace(num_layers1=layers, layer_name="GCNConv", num_features=K, hidden=H, num_classes=5)
    torch.manual_seed(K * 7 + H)
    model = network.Classify_node(args).cuda()
    with torch.no_grad():
        for c in model.conv:
            c.bias.normal_(std=0.1)
    x = torch.rand(n, K) - 0.3
    y = torch.randint(0, 5, (n,))
    tm = torch.rand(n) < 0.4
    tm[0] = True
    masks = None
    if mode == "eval":
        model.eval()
    else:
        model.train()
        if mode == "masks":
            masks = [(torch.rand(n, H) > 0.5).to(torch.uint8) for _ in range(layers)]
            model._inject_masks = [m.cuda() for m in masks]
    xg, eig = x.cuda(), ei.cuda()
    res = {}
    for narrow in (True, False):
        model.set_op_config(ops.DEFAULT.replace(narrow_input_first=narrow))
        model.zero_grad()
        torch.manual_seed(99)   # the hashed dropout draws its per-layer seeds from torch's generator
        # embed() is the graph-level models' path; the node-level head follows it here
        out = torch.nn.functional.log_softmax(model.head(model.embed(xg, eig)), dim=1)
        loss = torch.nn.functional.nll_loss(out[tm.cuda()], y.cuda()[tm.cuda()])
        loss.backward()
        res[narrow] = (out.detach().cpu(), float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()})
    g = model.conv[0].graph(eig, n)
    assert getattr(g, "_agg_input", None) is not None and g._agg_input[0] is xg, "the aggregate-first node did not run"
    assert g._agg_input[2].shape == (n, K)
    a, b = res[True], res[False]
    assert rel(a[0], b[0]) < 2e-5
    assert abs(a[1] - b[1]) < 2e-5 * abs(b[1])
    for k in b[2]:
        assert rel(a[2][k], b[2][k]) < 2e-4, k
    if mode != "hashed":
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, x, ei, y, num_layers=layers, train_mask=tm, masks=masks)
        assert rel(a[0], o_ref) < 1e-4
        for k in g_ref:
            assert rel(a[2][k], g_ref[k]) < 1e-3, k
    # an in-place edit of the input re-forms A_hat x
    before = g._agg_input[2].clone()
    xg.mul_(2.0)
    model.set_op_config(ops.DEFAULT)
    model.embed(xg, eig)
    assert rel(g._agg_input[2].cpu(), (2.0 * before).cpu()) < 1e-6


@pytest.mark.parametrize("F_,C,with_rows", [(512, 1, True), (64, 3, False), (256, 8, True), (16, 2, True), (1024, 1, False)])
def test_mean_pool_head_equals_pool_then_head(mods, F_, C, with_rows):
    """ops.MeanPoolHead (fitgnn_pool_head_f32 / _bwd_f32: lt1(global_mean_pool(x[rows])) in one launch each way) against the pool
    followed by the head in fp64: outputs, the input gradient on every row, the head's weight and bias gradients; empty graphs."""
    from fitgnn_amd import ops

    torch.manual_seed(F_ + C)
    G = 37
    sizes = torch.randint(0, 40, (G,))
    sizes[5] = 0
    sizes[-1] = 0
    batch_all = torch.repeat_interleave(torch.arange(G), sizes)
    n = int(batch_all.numel())
    x = torch.randn(n, F_)
    if with_rows:
        rows = torch.nonzero(torch.rand(n) < 0.6).flatten()
        batch = batch_all[rows]
    else:
        rows, batch = None, batch_all
    W, b = torch.randn(C, F_) / F_ ** 0.5, torch.randn(C)
    gy = torch.randn(G, C)
    xg = x.cuda().requires_grad_(True)
    Wg, bg = W.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    assert ops.pool_head_supported(xg, Wg)
    pi = ops.pool_index(batch.cuda(), G, None if rows is None else rows.cuda(), n)
    y = ops.MeanPoolHead.apply(xg, pi, Wg, bg, ops.DEFAULT)
    y.backward(gy.cuda())
    xr, Wr, br = x.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    sel = xr if rows is None else xr[rows]
    pooled = torch.zeros(G, F_, dtype=torch.float64).index_add_(0, batch, sel) / torch.bincount(batch, minlength=G).clamp(min=1).unsqueeze(1)
    ref = pooled @ Wr.t() + br
    ref.backward(gy.double())
    assert rel(y.detach().cpu().double(), ref.detach()) < 1e-5
    assert rel(xg.grad.cpu().double(), xr.grad) < 1e-5
    assert rel(Wg.grad.cpu().double(), Wr.grad) < 1e-5
    assert rel(bg.grad.cpu().double(), br.grad) < 1e-5
    # twice to the same bits
    xg2 = x.cuda().requires_grad_(True)
    y2 = ops.MeanPoolHead.apply(xg2, pi, Wg.detach(), bg.detach(), ops.DEFAULT)
    assert torch.equal(y2.detach(), y.detach())


def test_adam_folds_a_fresh_gradient_buffer_and_advances_its_state(mods):
    """FlatGrads.enable_fresh + FlatAdam.step (fitgnn_adam_step_acc_f32): gradients written to the fresh buffer (ops.OpConfig.grad_sink)
    give the weights, moments and accumulated gradients of the same gradients added tensor by tensor (bit for bit); the buffer is
    cleared; the step count and the seed bank move on once per step."""
    from fitgnn_amd import ops, train

    torch.manual_seed(0)
    shapes = [(64, 24), (64,), (64, 64), (64,), (1, 64), (1,)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    f1, f2 = train.FlatGrads(ps).enable_fresh(), train.FlatGrads(qs)
    o1, o2 = train.FlatAdam(f1, lr=0.01, weight_decay=5e-4), train.FlatAdam(f2, lr=0.01, weight_decay=5e-4)
    bank = ops.SeedBank(3, torch.device("cuda"))
    expect = bank.seeds.clone()
    o1.seed_bank = bank
    for it in range(4):
        for k, (p, q) in enumerate(zip(ps, qs)):
            g = torch.randn_like(q)
            if k % 2 == 0 or it == 2:
                f1.view(p.data_ptr()).copy_(g)     # a backward node that knows the sink
            else:
                p.grad.add_(g)                      # autograd's own accumulation
            q.grad.add_(g)                          # (never cleared between steps: run.py:254-304)
        o1.step(); o2.step()
        assert torch.equal(o1.P, o2.P) and torch.equal(o1.m, o2.m) and torch.equal(o1.v, o2.v), it
        assert torch.equal(f1.grads, f2.grads)
        assert float(f1.fresh.abs().max()) == 0.0
        assert float(o1.step_count[0]) == it + 1 and float(o1.step_count[1]) == 0.0
        expect.add_(ops.SeedBank.GOLD)   # (wraps modulo 2^64, as SeedBank.advance)
        assert torch.equal(bank.seeds, expect)
    assert f1.view(12345) is None


def test_padded_batch_plan_assembles_the_batch_of_any_graphs(mods):
    """graph_data.PaddedBatchPlan (fitgnn_batch_offsets / fitgnn_batch_gather): the batch of 16 arbitrary graphs assembled on the device
    from the dataset's global arrays == the batch GraphSet.batch_ids builds on the host (normalised CSR, pooled rows, first layer's
    aggregated input, targets: bit for bit), its tiles partition the padded row range, two consecutive steps walk the permutation."""
    from fitgnn_amd import graph_data, ops
    from fitgnn_amd.csr import csr_for

    mol = graph_data.synthetic_molecules(96, seed=9)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    B = 16
    plan = graph_data.PaddedBatchPlan(gset, B, lambda y: y[:, 2:3].long().float())
    rng = np.random.default_rng(1)
    ids = rng.permutation(96)[:2 * B]
    assert plan.fits(ids.reshape(2, B)).all()
    plan.set_epoch(ids)
    for step in range(2):
        plan.assemble()
        torch.cuda.synchronize()
        assert int(plan.step_idx) == step + 1
        want = gset.batch_ids(ids[step * B:(step + 1) * B].tolist(), "gs")
        g = csr_for(want["edge_index"], int(want["x"].shape[0]), "gcn")
        n, nnz = g.n, g.nnz
        assert n <= plan.R_cap and nnz <= plan.E_cap
        assert torch.equal(plan.b_rowptr[:n + 1], g.f.rowptr) and bool((plan.b_rowptr[n:] == nnz).all())
        assert torch.equal(plan.b_col[:nnz], g.f.col) and torch.equal(plan.b_val[:nnz], g.f.val)
        assert torch.equal(g.f.val, g.t.val) and torch.equal(g.f.col, g.t.col)          # (what lets the batch's transpose be itself)
        mask_idx = torch.nonzero(want["mask"]).flatten()
        m = int(mask_idx.numel())
        assert torch.equal(plan.b_members[:m].long(), mask_idx)
        graph_of = want["graph"][want["mask"]]
        assert torch.equal(plan.b_seg_of_row[:n][mask_idx].long(), graph_of) and int((plan.b_seg_of_row[:n] >= 0).sum()) == m
        assert bool((plan.b_seg_of_row[n:] == -1).all())
        # the compact view of the pooled rows (a last layer evaluated on them only): int64 row index, graph of a compact row, positions
        assert torch.equal(plan.b_members64[:m], mask_idx) and bool((plan.b_members64[m:] == 0).all())
        assert torch.equal(plan.b_cseg[:m].long(), graph_of) and bool((plan.b_cseg[m:] == -1).all())
        assert torch.equal(plan.b_pos[:n][mask_idx].long(), torch.arange(m, device="cuda"))
        rest = torch.ones(plan.R_cap, dtype=torch.bool, device="cuda")
        rest[mask_idx] = False
        assert bool((plan.b_pos[rest] >= plan.M_cap).all()) and bool((plan.b_pos[rest] < plan.M_cap + ops.ZERO_ROWS).all())
        cnt = torch.bincount(graph_of, minlength=B)
        assert torch.equal(plan.b_seg_off.long(), torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(cnt, 0)]))
        assert torch.equal(plan.b_inv_cnt, 1.0 / cnt.clamp(min=1).float())
        ax = ops.spmm_graph(g, want["x"].float())
        assert torch.equal(plan.b_ax[:n], ax) and float(plan.b_ax[n:].abs().max() if n < plan.R_cap else 0.0) == 0.0
        assert torch.equal(plan.b_tgt, want["y"][:, 2:3].long().float())
        t = plan.b_tiles.cpu().numpy()
        live = t[t[:, 1] > t[:, 0]]
        order = np.argsort(live[:, 0])
        live = live[order]
        assert live[0, 0] == 0 and live[-1, 1] == plan.R_cap and np.array_equal(live[1:, 0], live[:-1, 1])      # a partition of the rows
        assert np.all(live[:, 1] - live[:, 0] <= 16) and np.array_equal(live[:, 2], live[:, 0]) and np.array_equal(live[:, 3], live[:, 1] - live[:, 0])
        rp = plan.b_rowptr.cpu().numpy()
        assert np.array_equal(live[:, 4], rp[live[:, 0]]) and np.array_equal(live[:, 5], rp[live[:, 1]])
        # a product over the static batch == the product over the host-built one; the rows past the batch are written (zeros)
        X = torch.randn(plan.R_cap, 64, device="cuda")
        Y = ops.spmm_graph(plan.graph, X)
        assert torch.equal(Y[:n], ops.spmm_graph(g, X[:n].contiguous())) and float(Y[n:].abs().max() if n < plan.R_cap else 0.0) == 0.0


@pytest.mark.parametrize("kind,task,force_eager_every", [("gs", "graph_reg", 0), ("gs", "graph_reg", 2), ("gc", "graph_reg", 0), ("gs", "graph_cls", 0),
                                                        ("gc", "graph_cls", 3)])
def test_shuffled_epochs_replay_one_captured_step(mods, kind, task, force_eager_every):
    """GraphTrainer(reshuffle=True, capture=True): the batches of every epoch assembled on the device and stepped by ONE captured
    hipGraph == the eager rebuild of the same shuffled batches (same torch seed per epoch, dropout off): epoch losses and final weights;
    a last, shorter batch and batches declared not to fit take the eager way in between.  Subgraph and coarse views, regression (L1)
    and classification (max pool, softmax, cross-entropy)."""
    from fitgnn_amd import graph_data, train

    network, fnn, gorc = mods
    if task == "graph_reg":
        mol = graph_data.synthetic_molecules(110, seed=6)   # 110 = 6 x 16 + 14: a short last batch
        args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=64, num_classes=1)
        cls = network.Regress_graph_gs if kind == "gs" else network.Regress_graph_gc
        kw = dict(prop=1)
    else:
        mol = graph_data.synthetic_graph_classes(110, seed=6)
        args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=3, hidden=64, num_classes=2)
        cls = network.Classify_graph_gs if kind == "gs" else network.Classify_graph_gc
        kw = dict(task="graph_cls", multi_prop=False)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    torch.manual_seed(3)
    m1, m2 = cls(args).cuda(), cls(args).cuda()
    m2.load_state_dict(m1.state_dict())
    m1.dropout_p = m2.dropout_p = 0.0
    t1 = train.GraphTrainer(m1, gset, list(range(110)), kind=kind, batch_size=16, reshuffle=True, **kw)
    t2 = train.GraphTrainer(m2, gset, list(range(110)), kind=kind, batch_size=16, reshuffle=True, capture=True, **kw)
    assert t1._plan is None and t2._plan is not None and t2._plan.kind == kind
    t2.steps_per_graph = 4   # six full batches per epoch: one captured run of four steps, then two single steps
    if force_eager_every:
        t2._plan.fits = lambda ids: np.arange(len(ids)) % force_eager_every == 0
    for epoch in range(3):
        torch.manual_seed(100 + epoch)
        a = float(t1.step())
        torch.manual_seed(100 + epoch)
        b = float(t2.step())
        assert a == pytest.approx(b, rel=2e-5), (epoch, a, b)
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert rel(w, v) < 2e-4, k
    # dropout on: the replayed step draws fresh patterns (the optimiser kernel moves the seeds on)
    m2.dropout_p = 0.5
    torch.manual_seed(7)
    l1 = float(t2.step())
    torch.manual_seed(7)
    l2 = float(t2.step())
    assert l1 != l2
    # "auto" on a model the plan does not take (attention layers): the order is drawn once, the captured per-batch steps stay
    if kind == "gs" and task == "graph_reg" and not force_eager_every:
        gargs = argparse.Namespace(num_layers1=2, layer_name="GATConv", num_features=11, hidden=64, num_classes=1)
        t4 = train.GraphTrainer(network.Regress_graph_gs(gargs).cuda(), gset, list(range(64)), kind="gs", batch_size=16, prop=1, reshuffle="auto",
                                capture=True)
        assert t4._plan is None and t4._rebuild is None and t4.capture
        t5 = train.GraphTrainer(m1, gset, list(range(64)), kind="gs", batch_size=16, prop=1, reshuffle="auto", capture=True)
        assert t5._plan is not None and t5._rebuild is not None


@pytest.mark.parametrize("hidden,classes", [(32, 4), (64, 47)])
def test_gd_step_replayed_from_a_hipgraph_equals_the_eager_step(mods, hidden, classes):
    """GDTrainer's lean, captured step (weight gradients to the optimiser's fresh buffer, ONE Adam launch, the whole step replayed
    from a hipGraph) against the plain eager step (lean_step=False, capture=False): losses and weights over four epochs bit for bit
    (dropout off), the gradients it leaves in p.grad; a step run under profiling hooks takes the eager path and still matches; with
    dropout on successive replays draw different patterns."""
    from fitgnn_amd import train

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=3)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=hidden, num_classes=classes)
    if classes > 4:
        batch.y = torch.randint(0, classes, batch.y.shape, device=batch.y.device)
    torch.manual_seed(5)
    m1, m2 = network.Classify_node(args).cuda(), network.Classify_node(args).cuda()
    m2.load_state_dict(m1.state_dict())
    m1.dropout_p = m2.dropout_p = 0.0
    t1 = train.GDTrainer(m1, batch, lr=0.01, weight_decay=5e-4, lean_step=False, capture=False)
    t2 = train.GDTrainer(m2, batch, lr=0.01, weight_decay=5e-4)
    assert not t1.capture and t1.cfg.grad_sink is None and t2.capture and t2.cfg.grad_sink is t2.flat
    for epoch in range(4):
        if epoch == 2:
            t2.cfg.profile = []        # hooks on: this step runs eagerly (and records its SpMM launches)
        a, b = float(t1.step()), float(t2.step())
        if epoch == 2:
            assert len(t2.cfg.profile) >= 3
            t2.cfg.profile = None
        assert a == b, (epoch, a, b)
        for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p.detach(), q.detach()), (epoch, k)
            assert torch.equal(p.grad, q.grad), (epoch, k)
    assert t2._graph is not None
    m2.dropout_p = 0.5
    t3 = train.GDTrainer(m2, batch, lr=0.0, weight_decay=0.0, capture=True)
    l1 = float(t3.step())
    l2 = float(t3.step())
    assert l1 != l2


def test_a_stale_autograd_graph_makes_the_trainers_step_eagerly_instead_of_capturing(mods):
    """A `loss` of an earlier eager training phase that is still referenced keeps the parameters' AccumulateGrad nodes bound to the stream
    that phase ran on; capturing a step on another stream would fork that stream into the hipGraph without a join (main.py --exp_setup
    Gc_train_2_Gs_train did that once: the process died ending the capture).  train._accumulate_stream_guard sees autograd's report of it
    during the warm-up steps and the trainers fall back to eager steps; once the reference is gone they capture."""
    from fitgnn_amd import train

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=4)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=24, hidden=32, num_classes=4)
    torch.manual_seed(5)
    model = network.Classify_node(args).cuda()
    model.dropout_p = 0.0
    out = model(batch.x, batch.edge_index)
    stale = torch.nn.functional.nll_loss(out.index_select(0, batch.train_idx), batch.y.index_select(0, batch.train_idx))
    stale.backward()                       # ... and `stale` stays alive
    ref = train.GDTrainer(model, batch, lr=0.0, weight_decay=0.0, lean_step=False, capture=False)
    want = float(ref.step())
    tr = train.GDTrainer(model, batch, lr=0.0, weight_decay=0.0, capture=True)
    got = float(tr.step())
    assert not tr.capture and tr._graph is None and got == want
    mb = train.MBTrainer(model, batch, batch_size=8, lr=0.0, weight_decay=0.0, capture=True)
    mb.step()
    assert not mb.capture and mb._graphs is None
    del stale, out
    tr2 = train.GDTrainer(model, batch, lr=0.0, weight_decay=0.0, capture=True)
    assert float(tr2.step()) == want and tr2.capture and tr2._graph is not None


@pytest.mark.parametrize("cls_name,layers", [("Regress_graph_gs", 2), ("Regress_graph_gs", 1), ("Regress_graph_gs", 3), ("Classify_graph_gs", 2)])
@pytest.mark.parametrize("mode", ["eval", "masks", "hashed"])
def test_last_layer_on_the_pooled_rows_changes_nothing_the_pool_sees(mods, cls_name, layers, mode):
    """ops.FusedGCNLayerRows behind the *_graph_gs models (the last GCN layer aggregate-first on x[mask]'s rows, compact output, the
    pool over the compact rows) against the layer over every row (OpConfig(pooled_rows_last_layer=False)): model output, loss and
    every gradient; eval, injected dropout masks, hashed dropout."""
    from fitgnn_amd import graph_data, ops, train
    import types

    network, fnn, gorc = mods
    reg = cls_name.startswith("Regress")
    mol = graph_data.synthetic_molecules(24, seed=8) if reg else graph_data.synthetic_graph_classes(24, seed=8)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    F_in = 11 if reg else 3
    args = argparse.Namespace(num_layers1=layers, layer_name="GCNConv", num_features=F_in, hidden=64, num_classes=1 if reg else 2)
    torch.manual_seed(layers + 17)
    model = getattr(network, cls_name)(args).cuda()
    with torch.no_grad():
        for c in model.conv:
            c.bias.normal_(std=0.1)
    b = train._cat_pieces([gset.batch(0, 24, "gs")], "gs", types)
    n = int(b["x"].shape[0])
    if mode == "eval":
        model.eval()
    else:
        model.train()
        if mode == "masks":
            model._inject_masks = [(torch.rand(n, 64, device="cuda") > 0.5).to(torch.uint8) for _ in range(layers)]
    w = torch.randn(24, 1 if reg else 2, device="cuda")
    res = {}
    for flag in (True, False):
        model.set_op_config(ops.DEFAULT.replace(pooled_rows_last_layer=flag))
        model.zero_grad()
        torch.manual_seed(5)
        out = model(b, b["graph_of_masked"])
        (out * w).sum().backward()
        res[flag] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    assert int(b["mask_idx"].numel()) < n          # (the pooled rows are a proper subset: about half)
    assert rel(res[True][0], res[False][0]) < 3e-5
    for k in res[False][1]:
        assert rel(res[True][1][k], res[False][1][k]) < 3e-4, k


def test_appnp_trainer_takes_the_loss_on_the_train_rows_only(mods):
    """GDTrainer on network.APPNPNet: softmax + NLL on the train rows of the model's logits (ops.SoftmaxNLL) == log_softmax over every
    row followed by NLLLoss on out[mask] (the model's forward): losses and weights over three steps (dropout off)."""
    from fitgnn_amd import train

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=6)
    args = argparse.Namespace(num_features=24, hidden=32, num_classes=4, K=4, alpha=0.1, dropout=0.0)
    torch.manual_seed(9)
    m1, m2 = network.APPNPNet(args).cuda(), network.APPNPNet(args).cuda()
    m2.load_state_dict(m1.state_dict())
    t1, t2 = train.GDTrainer(m1, batch, lr=0.01, weight_decay=5e-4), train.GDTrainer(m2, batch, lr=0.01, weight_decay=5e-4)
    assert t1.fused_logits and t2.fused_logits
    t2.fused_logits = False
    for step in range(3):
        a, b = float(t1.step()), float(t2.step())
        assert a == pytest.approx(b, rel=1e-5), (step, a, b)
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert rel(v, w) < 1e-4, k


def test_appnp_step_replayed_from_a_hipgraph_equals_the_eager_step(mods):
    """GDTrainer(capture="auto") on network.APPNPNet (a small union): forward, loss on the padded view, backward and Adam replayed from
    one hipGraph == the eager step, bit for bit (dropout off: torch's generator is the only difference otherwise), over five steps;
    with dropout on the replays draw fresh masks (the losses differ from step to step and stay finite)."""
    from fitgnn_amd import train

    network, fnn, gorc = mods
    batch, _ = _subgraph_batches(seed=8)
    args = argparse.Namespace(num_features=24, hidden=32, num_classes=4, K=5, alpha=0.1, dropout=0.0)
    torch.manual_seed(11)
    m1, m2 = network.APPNPNet(args).cuda(), network.APPNPNet(args).cuda()
    m2.load_state_dict(m1.state_dict())
    t1 = train.GDTrainer(m1, batch, lr=0.01, weight_decay=5e-4, capture="auto")
    t2 = train.GDTrainer(m2, batch, lr=0.01, weight_decay=5e-4, capture=False)
    assert t1.capture and not t2.capture
    for step in range(5):
        a, b = float(t1.step()), float(t2.step())
        assert a == b, (step, a, b)
    assert t1._graph is not None
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(v, w), k
    args.dropout = 0.5
    m3 = network.APPNPNet(args).cuda()
    t3 = train.GDTrainer(m3, batch, lr=0.0, weight_decay=0.0, capture="auto")   # lr 0: the weights stay, only the masks change
    losses = [float(t3.step()) for _ in range(4)]
    assert t3._graph is not None and all(np.isfinite(losses)) and len(set(losses)) == 4, losses


@pytest.mark.parametrize("C", [3, 47, 64])
@pytest.mark.parametrize("sizes", [[100, 7, 17, 300, 3, 3, 64, 33, 2, 5, 5, 40, 65, 1, 800, 769, 768], [5] * 200, [70, 200], [1500, 1025]])
def test_appnp_in_lds_equals_the_per_step_propagation(mods, C, sizes):
    """fitgnn_appnp_units_f32 (the K steps of a subgraph of <= 64 rows between two LDS buffers, one launch) + the per-step kernel on the
    sub-matrix of the larger subgraphs == the per-step kernel over every row (OpConfig(appnp_in_lds=False)): propagated signal and the
    gradient w.r.t. z_0; blocks around the unit size, tiny blocks packed into one unit, a batch with no unit at all."""
    from fitgnn_amd import csr, ops

    rng = np.random.default_rng(C + len(sizes))
    src, dst, off = [], [], 0
    for sz in sizes:   # every block: a ring plus chords (connected, symmetric)
        ring = np.arange(sz)
        und = {(min(a, b), max(a, b)) for a, b in zip(ring, np.roll(ring, -1)) if a != b}
        for _ in range(sz // 2):
            a, b = rng.integers(0, sz, size=2)
            if a != b:
                und.add((min(a, b), max(a, b)))
        if und:
            u = np.array(sorted(und), dtype=np.int64) + off
            src += [u[:, 0], u[:, 1]]; dst += [u[:, 1], u[:, 0]]
        off += sz
    ei = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)])).cuda()
    n = off
    g = csr.CSRGraph(ei, n, mode="gcn")
    h4 = (C + 3) // 4
    plan = ops.appnp_plan(g, h4)
    assert plan.cap_rows == min(768, 768 // h4)
    rp = g.f.rowptr.cpu().numpy()
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    L = _lib_mod().lib()
    small = sum(int(sz) for sz, a, b in zip(sizes, bounds[:-1], bounds[1:])
                if sz <= plan.cap_rows and rp[b] - rp[a] <= L.fitgnn_appnp_unit_entries())
    # the larger blocks: all of these fit LDS one slice at a time (the column-sliced kernel, sixteen wavefronts each)
    assert plan.rows_in_units == small and plan.rows_in_lds_blocks == n - small and plan.n_open == 0
    old = ops.appnp_plan(g, h4, sliced=False)
    assert old.rows_in_units == small and old.n_open == n - small and old.n_lds_blocks == 0
    assert plan.max_rows <= plan.cap_rows and (plan.n_units == 0 or plan.max_rows >= min(max(sizes), 1))
    z0 = torch.randn(n, C, device="cuda")
    w = torch.randn(n, C, device="cuda")
    res = {}
    for flag, cfg in (("sliced", ops.DEFAULT), ("whole", ops.DEFAULT.replace(appnp_sliced=False)), (False, ops.DEFAULT.replace(appnp_in_lds=False))):
        zz = z0.clone().requires_grad_(True)
        out = ops.APPNPPropagate.apply(zz, g, 10, 0.1, cfg)
        (out * w).sum().backward()
        res[flag] = (out.detach(), zz.grad.detach())
    for flag in ("sliced", "whole"):
        assert rel(res[flag][0], res[False][0]) < 1e-5, flag
        assert rel(res[flag][1], res[False][1]) < 1e-5, flag
    # K = 1 and a different alpha
    a = ops.APPNPPropagate.apply(z0, g, 1, 0.3, ops.DEFAULT)
    b = ops.APPNPPropagate.apply(z0, g, 1, 0.3, ops.DEFAULT.replace(appnp_in_lds=False))
    assert rel(a, b) < 1e-6


def _ring_blocks(sizes, rng, hub_every=0):
    """Block-diagonal symmetric pattern: every block a ring plus chords; hub_every > 0: every hub_every-th node of a block is also tied
    to the block's first node (a long row, as a star's centre)."""
    src, dst, off = [], [], 0
    for sz in sizes:
        sz = int(sz)
        ring = np.arange(sz)
        und = {(min(a, b), max(a, b)) for a, b in zip(ring, np.roll(ring, -1)) if a != b}
        for _ in range(sz // 2):
            a, b = rng.integers(0, sz, size=2)
            if a != b:
                und.add((min(a, b), max(a, b)))
        if hub_every:
            und |= {(0, int(j)) for j in range(hub_every, sz, hub_every)}
        if und:
            u = np.array(sorted(und), dtype=np.int64) + off
            src += [u[:, 0], u[:, 1]]; dst += [u[:, 1], u[:, 0]]
        off += sz
    return torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)])).cuda(), off


@pytest.mark.parametrize("C,K", [(3, 10), (47, 10), (47, 1), (47, 3), (64, 2)])
def test_appnp_one_workgroup_per_large_subgraph_equals_the_per_step_propagation(mods, C, K):
    """fitgnn_appnp_blocks_f32: the subgraphs beyond a unit, one workgroup each, all K steps in one launch between two scratch signals
    (CSR slice in LDS) == the per-step kernel, BIT FOR BIT on those rows (same row arithmetic), forward and the gradient w.r.t. z_0;
    odd and even K (which scratch signal the last step reads), long rows (split over the wave's slots), a block beyond the row
    capacity and small blocks beside them (those stay on the per-step kernel / the units)."""
    from fitgnn_amd import csr, ops

    L = _lib_mod().lib()
    h4 = (C + 3) // 4
    cap = int(L.fitgnn_appnp_unit_rows(h4))
    rng = np.random.default_rng(100 * C + K)
    sizes = [int(v) for v in rng.integers(cap + 1, cap + 500, size=ops.AppnpPlan.MIN_BLOCKS + 10)] + [5] * 20 + [int(L.fitgnn_appnp_block_rows()) + 1, 7, cap + 3]
    ei, n = _ring_blocks(sizes, rng, hub_every=3)
    g = csr.CSRGraph(ei, n, mode="gcn")
    plan = ops.appnp_plan(g, h4, sliced=False)
    big = [sz for sz in sizes if cap < sz <= L.fitgnn_appnp_block_rows()]
    assert plan.n_blocks == len(big) and plan.rows_in_blocks == sum(big) and plan.block_max_rows == max(big)
    assert plan.n_open == int(L.fitgnn_appnp_block_rows()) + 1 and plan.rows_in_units == n - plan.n_open - plan.rows_in_blocks
    assert ops.appnp_plan(g, h4, blocks=False, sliced=False).n_blocks == 0
    sl = ops.appnp_plan(g, h4)   # the default: these blocks fit LDS a slice at a time, the workgroup-per-block launch has nothing left
    assert sl.n_lds_blocks == len(big) and sl.n_blocks == 0 and sl.n_open == plan.n_open
    assert sum(la[2] for la in sl.lds_launches) == len(big)
    assert all(L.fitgnn_appnp_lds_bytes(r, e, w) <= L.fitgnn_appnp_lds_max_bytes() and r * w <= 4 * t for w, _, _, r, e, t in sl.lds_launches)
    in_blocks = torch.zeros(n, dtype=torch.bool, device="cuda")
    for a, b in plan.blocks.cpu().numpy():
        in_blocks[a:b] = True
    z0 = torch.randn(n, C, device="cuda")
    w = torch.randn(n, C, device="cuda")
    res = {}
    whole = ops.DEFAULT.replace(appnp_sliced=False)
    for name, cfg in (("blocks", whole), ("steps", ops.DEFAULT.replace(appnp_in_lds=False)), ("no_blocks", whole.replace(appnp_blocks=False)),
                      ("sliced", ops.DEFAULT)):
        zz = z0.clone().requires_grad_(True)
        out = ops.APPNPPropagate.apply(zz, g, K, 0.1, cfg)
        (out * w).sum().backward()
        res[name] = (out.detach(), zz.grad.detach())
    for i in (0, 1):
        assert torch.equal(res["blocks"][i][in_blocks], res["steps"][i][in_blocks]), ("rows in blocks", i)
        assert torch.equal(res["blocks"][i], res["no_blocks"][i]), ("against the sub-matrix path", i)
        assert rel(res["blocks"][i], res["steps"][i]) < 1e-5
        assert rel(res["sliced"][i], res["steps"][i]) < 1e-5
        assert rel(res["sliced"][i][in_blocks], res["steps"][i][in_blocks]) < 1e-5


@pytest.mark.parametrize("C,threads,slice_", [(3, 64, 1), (10, 128, 4), (10, 64, 2), (47, 256, 4), (47, 1024, 1), (47, 512, 2), (50, 192, 4), (64, 1024, 4)])
def test_appnp_column_sliced_lds_kernel_for_every_launch_shape(mods, C, threads, slice_):
    """fitgnn_appnp_lds_f32 called directly: every block its own range, workgroups of 1 .. 16 wavefronts, slices of 1 / 2 / 4 float4 columns
    (h4 = 3 at slice 4 = passes of 2 + 1 columns, h4 = 13 = 4 + 4 + 4 + 1), rows longer than 16 entries (a wavefront's: 1, 2 and many
    rounds of 64 / w entries) next to short ones, K odd and even == the per-step kernel, forward and the adjoint; and the argument checks."""
    from fitgnn_amd import csr, ops

    L = _lib_mod().lib()
    h4 = (C + 3) // 4
    cap = 4 * threads // slice_
    while L.fitgnn_appnp_lds_bytes(cap, 5 * cap, slice_) > L.fitgnn_appnp_lds_max_bytes():   # (about four entries per row below)
        cap = cap * 3 // 4
    rng = np.random.default_rng(C + threads + slice_)
    sizes = [int(v) for v in rng.integers(1, cap + 1, size=40)] + [cap, 1, 2, cap]
    ei, n = _ring_blocks(sizes, rng, hub_every=2)
    g = csr.CSRGraph(ei, n, mode="gcn")
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ranges = torch.from_numpy(np.stack([bounds[:-1], bounds[1:]], 1).copy()).cuda()
    rp = g.f.rowptr.cpu().numpy()
    max_e = int((rp[bounds[1:]] - rp[bounds[:-1]]).max())
    assert L.fitgnn_appnp_lds_bytes(cap, max_e, slice_) <= L.fitgnn_appnp_lds_max_bytes()
    z0 = torch.randn(n, C, device="cuda")
    w = torch.randn(n, C, device="cuda")
    pad = lambda t: ops.APPNPPropagate._padded(t.contiguous(), h4)   # noqa: E731
    st = _lib_mod().stream_ptr(z0.device)
    dp = _lib_mod().dptr
    for K in (1, 4, 7):
        zz = z0.clone().requires_grad_(True)
        ref = ops.APPNPPropagate.apply(zz, g, K, 0.15, ops.DEFAULT.replace(appnp_in_lds=False))
        (ref * w).sum().backward()
        for side, x, want, bwd in ((g.f, pad(z0), ref.detach(), 0), (g.t, pad(w), zz.grad, 1)):
            y = torch.full_like(x, float("nan"))
            rc = L.fitgnn_appnp_lds_f32(dp(side.rowptr), dp(side.col), dp(side.val), dp(ranges), len(sizes), cap, max_e, dp(x), dp(y), h4, K, 0.15,
                                        bwd, threads, slice_, st)
            assert rc == 0
            assert rel(y[:, :C], want) < 1e-5, (K, bwd)
            assert not torch.isnan(y).any() and (C == 4 * h4 or float(y[:, C:].abs().max()) == 0.0)
    x = pad(z0)
    y = torch.empty_like(x)
    args = lambda **kw: [dp(g.f.rowptr), dp(g.f.col), dp(g.f.val), dp(ranges), len(sizes), kw.get("rows", cap), max_e, dp(x), dp(y), h4, 3, 0.1, 0,  # noqa: E731
                         kw.get("threads", threads), kw.get("slice_", slice_), st]
    assert L.fitgnn_appnp_lds_f32(*args(rows=4 * threads // slice_ + 1)) == -1          # more items than four per thread
    assert L.fitgnn_appnp_lds_f32(*args(threads=threads + 1)) == -1   # whole wavefronts
    assert L.fitgnn_appnp_lds_f32(*args(slice_=3)) == -1


@pytest.mark.parametrize("C", [3, 47, 48])
def test_appnp_takes_the_table_rows_and_hands_the_padded_signal_to_the_loss(mods, C):
    """APPNPPropagate(z_table, ..., row_index): the de-duplicated table's rows gathered straight into the padded layout
    (fitgnn_gather_rows_padded_f32) and summed back per table row in backward (fitgnn_segment_sum_f32) == index_select + propagate;
    the result is the [rows x C] view of the padded signal, which softmax_nll_raw reads in place and answers with an equally strided
    gradient that backward takes as it is == SoftmaxNLL on a contiguous copy (loss, gradient w.r.t. the table)."""
    from fitgnn_amd import csr, ops

    rng = np.random.default_rng(C)
    sizes = [int(v) for v in rng.integers(1, 300, size=60)]
    ei, n = _ring_blocks(sizes, rng, hub_every=5)
    g = csr.CSRGraph(ei, n, mode="gcn")
    n_table = 700
    idx = torch.from_numpy(rng.integers(0, n_table, size=n)).cuda()
    ri = ops.RowIndex(idx, n_table)
    zt = torch.randn(n_table, C, device="cuda")
    w = torch.randn(n, C, device="cuda")
    a_in = zt.clone().requires_grad_(True)
    a = ops.APPNPPropagate.apply(a_in, g, 6, 0.1, ops.DEFAULT, ri)
    assert a.shape == (n, C) and a.stride() == (4 * ((C + 3) // 4), 1)
    (a * w).sum().backward()
    b_in = zt.clone().requires_grad_(True)
    b = ops.APPNPPropagate.apply(b_in.index_select(0, idx), g, 6, 0.1, ops.DEFAULT.replace(appnp_in_lds=False))
    (b * w).sum().backward()
    assert rel(a, b) < 1e-5 and rel(a_in.grad, b_in.grad) < 1e-5
    # the loss on the view
    train = torch.from_numpy(rng.choice(n, size=n // 3, replace=False)).cuda()
    y = torch.from_numpy(rng.integers(0, C, size=train.numel())).cuda()
    c_in = zt.clone().requires_grad_(True)
    z = ops.APPNPPropagate.apply(c_in, g, 6, 0.1, ops.DEFAULT, ri)
    loss, dz = ops.softmax_nll_raw(z, train, y, 1.0 / train.numel())
    assert dz.shape == (n, C) and dz.stride() == z.stride()
    z.backward(dz)
    d_in = zt.clone().requires_grad_(True)
    z2 = ops.APPNPPropagate.apply(d_in, g, 6, 0.1, ops.DEFAULT, ri).contiguous()
    loss2 = ops.SoftmaxNLL.apply(z2, train, y, 1.0 / train.numel())
    loss2.backward()
    assert float(loss[0]) == pytest.approx(float(loss2), rel=1e-6)
    assert rel(c_in.grad, d_in.grad) < 1e-6
    want = torch.nn.functional.nll_loss(torch.log_softmax(b.detach().double(), 1)[train], y, reduction="mean")
    assert float(loss[0]) == pytest.approx(float(want), rel=1e-4)


def test_csr_row_sum_with_eight_lanes_per_row(mods):
    """fitgnn_csr_row_sum_f32 (GAT's d a_src on the transposed edge order): rows of 0, 1, 7, 8, 9, 16, 17, 300 and 5 000 entries, a
    row count that is not a multiple of eight == float64 sums (<= 1e-6 relative), and the same bits on a second call."""
    L = _lib_mod().lib()
    dp = _lib_mod().dptr
    rng = np.random.default_rng(4)
    lens = np.array([0, 1, 7, 8, 9, 16, 17, 300, 0, 5000, 3, 2, 4] + list(rng.integers(0, 40, size=1000)), dtype=np.int64)
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    v = torch.randn(int(lens.sum()), device="cuda")
    n = len(lens)
    y = torch.full((n,), float("nan"), device="cuda")
    st = _lib_mod().stream_ptr(v.device)
    assert L.fitgnn_csr_row_sum_f32(dp(rowptr), dp(v), n, dp(y), st) == 0
    want = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, torch.repeat_interleave(torch.arange(n, device="cuda"), torch.from_numpy(lens).cuda()), v.double())
    assert not torch.isnan(y).any()
    assert float((y.double() - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1)
    y2 = torch.empty_like(y)
    assert L.fitgnn_csr_row_sum_f32(dp(rowptr), dp(v), n, dp(y2), st) == 0 and torch.equal(y, y2)


def test_appnp_few_large_subgraphs_stay_on_the_per_step_kernel(mods):
    """Fewer blocks than AppnpPlan.MIN_BLOCKS: no workgroup-per-block launch (it would leave most of the chip idle)."""
    from fitgnn_amd import csr, ops

    rng = np.random.default_rng(3)
    ei, n = _ring_blocks([300, 400, 5, 5], rng)
    g = csr.CSRGraph(ei, n, mode="gcn")
    plan = ops.appnp_plan(g, 12, sliced=False)
    assert plan.n_blocks == 0 and plan.blocks is None and plan.n_open == 700
    assert ops.appnp_plan(g, 12).n_open == 0   # (LDS holds them a slice at a time: no threshold there)
