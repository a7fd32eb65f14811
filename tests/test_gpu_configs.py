"""GPU tier: BASELINE.json's configurations at their full sizes through the path bench.py times -- GDTrainer.step's own call
(train.py:189: de-duplicated layer-0 table, embed_and_head(loss_rows=train_idx, compact_logits=True) -> aggregate-first last
layer on the loss rows, compact dZ, fused-derivative backward SpMM, fused loss; hidden 512); _fast_path asserts through the
launch log that those launches ran.

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

CONFIGS = ["S-pubmed", "S-physics", "S-products"]   # node-level configurations; S-qm9 (graph level) below


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


def _fast_path(model, batch, scale, masks=None, **cfg_kw):
    """Forward + fused loss + backward EXACTLY as GDTrainer.step issues them (train.py:189): the de-duplicated layer-0 table, then
    embed_and_head(..., loss_rows=batch.train_idx, compact_logits=True) under the default OpConfig -> ops.FusedGCNLastLayerRows
    (aggregate-first last layer, the head on the loss rows, compact dZ, the backward SpMM with layer 0's ELU' / dropout' in its
    store -- on a segmented batch the two-hop launch that also carries layer 0's own backward SpMM), the fused softmax + NLL on the
    compact logits.  masks=None: eval mode (dropout off); a list of two uint8
    [rows, hidden] masks: training mode with those dropout patterns injected (the oracle takes the same ones).
    Returns (compact logits [len(train_idx), C] in the order of train_idx, loss, gradients, launch kinds seen by cfg.profile);
    asserts that the timed path really ran: compact logits, a 'compact_dz' / 'two_hop' backward launch and a 'table' / 'gather' layer-0
    launch -- a silent fall-back to the all-rows form fails here."""
    from fitgnn_amd import ops
    from fitgnn_amd.ops import SoftmaxNLL

    cfg = ops.OpConfig(profile=[], **cfg_kw)  # the default switches (what GDTrainer runs under) + the launch log
    model.set_op_config(cfg)
    if masks is None:
        model.eval()
        model._inject_masks = None
    else:
        model.train()
        model._inject_masks = masks
    model.zero_grad()
    idx = batch.train_idx
    z = model.embed_and_head(batch.x_table, batch.edge_index, batch.row_index, loss_rows=idx, compact_logits=True)
    assert z.shape[0] == idx.numel() and z.shape[0] != batch.n_rows, "the last layer did not run on the loss rows"
    ar = torch.arange(z.shape[0], dtype=torch.int64, device=z.device)
    loss = SoftmaxNLL.apply(z, ar, batch.y.index_select(0, idx), scale)
    loss.backward()
    torch.cuda.synchronize()
    kinds = [k for _, _, k in cfg.profile]
    assert "compact_dz" in kinds or "two_hop" in kinds, kinds   # the last layer's backward SpMM: compact operand + fused derivative
    assert any(k in ("table", "gather") for k in kinds), kinds   # layer 0 on the de-duplicated table
    model._inject_masks = None
    model.set_op_config(ops.DEFAULT)
    return z.detach(), float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, kinds


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
    """The step bench.py times (GDTrainer.step's call: see _fast_path) against the oracle, at hidden 512 and the configuration's
    class count: compact logits == the oracle's log-probabilities on the train rows (<= 1e-4), loss, every gradient (<= 1e-3),
    in eval mode on four loader slices and in training mode (injected dropout masks) on one."""
    from fitgnn_amd import data, workloads
    from oracle import gnn_oracle as gorc

    name, sub = cfg["name"], cfg["sub"]
    n_c = cfg["wl"]["n_clusters"]
    model = _model(name)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    full = workloads.batch_from_subgraphs(name, sub, cfg["dev"], cfg["X"], cfg["y"])
    scale = 1.0 / float(full.train_idx.numel())
    z_full, loss_full, g_full, kinds_full = _fast_path(model, full, scale)
    assert np.isfinite(loss_full)
    if name == "S-products":   # the union whose stars go to the whole-subgraph kernel: all four products of the step ran, the two
        # backward ones in the two-hop launch
        assert full.graph.f.blocks is not None and kinds_full.count("tile") == 1 and kinds_full.count("two_hop") == 1, kinds_full
        # ... and the two-hop pass gives the two separate launches' gradients at full size (8.2 M rows, a 5-GB side table): bit for bit,
        # layer 0's bias gradient to fp32 summation order
        z2, loss2, g2, kinds2 = _fast_path(model, full, scale, two_hop_backward=False)
        assert kinds2.count("tile") == 2 and "compact_dz" in kinds2 and "two_hop" not in kinds2, kinds2
        assert torch.equal(z2, z_full) and loss2 == loss_full
        for k in g_full:
            if k == "conv.0.bias":
                assert float((g2[k] - g_full[k]).abs().max()) <= 1e-5 * float(g_full[k].abs().max()) + 1e-12, k
            else:
                assert torch.equal(g2[k], g_full[k]), k
    full_idx = full.train_idx
    n_batches = (n_c + 127) // 128
    picks = sorted({0, n_batches // 3, (2 * n_batches) // 3, n_batches - 1})
    ptr = sub["ptr"].cpu().numpy()
    for b in picks:
        clusters = np.arange(b * 128, min((b + 1) * 128, n_c))
        part = workloads.batch_from_subgraphs(name, data.select_clusters(sub, clusters), cfg["dev"], cfg["X"], cfg["y"])
        z, loss, g, _ = _fast_path(model, part, scale)
        r0, r1 = int(ptr[clusters[0]]), int(ptr[clusters[-1] + 1])
        assert part.n_rows == r1 - r0
        # link 2: the full union's logits of this slice's train rows == the slice run
        lo, hi = (int(v) for v in torch.searchsorted(full_idx, torch.tensor([r0, r1], device=full_idx.device)))
        assert hi - lo == z.shape[0] and torch.equal(full_idx[lo:hi] - r0, part.train_idx)
        assert rel(z_full[lo:hi].cpu(), z.cpu()) < 2e-5, (name, b)
        # link 1: the slice run == the oracle
        tm = part.train_mask.cpu()
        o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, part.x.cpu(), part.edge_index.cpu(), part.y.cpu(), num_layers=2,
                                                         train_mask=tm, loss_scale=scale)
        assert rel(torch.log_softmax(z, 1).cpu(), o_ref[tm]) < 1e-4, (name, b)   # Classify_node's output (network.py:35) on out[mask] (run.py:193-204)
        assert abs(loss - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (name, b, loss, float(l_ref))
        for k in g:
            assert rel(g[k].cpu(), g_ref[k]) < 1e-3, (name, b, k)
        if b == picks[1]:   # the same slice in TRAINING mode: dropout patterns injected on both sides
            torch.manual_seed(17 + b)
            masks = [(torch.rand(part.n_rows, 512, device=cfg["dev"]) > 0.5).to(torch.uint8) for _ in range(2)]
            zt, losst, gt, kinds = _fast_path(model, part, scale, masks=masks)
            o_ref, l_ref, g_ref = gorc.classify_node_fwd_bwd(sd, part.x.cpu(), part.edge_index.cpu(), part.y.cpu(), num_layers=2,
                                                             train_mask=tm, masks=[m.cpu() for m in masks], loss_scale=scale)
            assert rel(torch.log_softmax(zt, 1).cpu(), o_ref[tm]) < 1e-4, (name, b, "train")
            assert abs(losst - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (name, b, losst, float(l_ref))
            for k in gt:
                assert rel(gt[k].cpu(), g_ref[k]) < 1e-3, (name, b, k, "train")
        del part


def test_gradient_is_additive_over_shards(cfg):
    """Full-size identity behind the data-parallel step: with the global 1 / count scale, the gradient of the whole union
    equals the sum of the gradients of its data.shard_clusters() shards."""
    from fitgnn_amd import data, workloads

    name, sub = cfg["name"], cfg["sub"]
    model = _model(name)
    full = workloads.batch_from_subgraphs(name, sub, cfg["dev"], cfg["X"], cfg["y"])
    scale = 1.0 / float(full.train_idx.numel())
    _, loss_full, g_full, _ = _fast_path(model, full, scale)
    del full
    world = 3
    owner = data.shard_clusters(None, cfg["nnz_c"], world)
    load = np.bincount(owner, weights=cfg["nnz_c"], minlength=world)
    assert load.max() - load.min() <= cfg["nnz_c"].max()
    acc, loss_sum = None, 0.0
    for k in range(world):
        part = workloads.batch_from_subgraphs(name, data.select_clusters(sub, np.nonzero(owner == k)[0]), cfg["dev"], cfg["X"], cfg["y"])
        _, loss, g, _ = _fast_path(model, part, scale)
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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json config 5: QM9 graph regression at its full size (S-qm9: 130 831 molecules of ~18 nodes)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def qm9():
    from fitgnn_amd import graph_data

    n = 130_831
    mol = graph_data.synthetic_molecules(n, seed=0)
    gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
    return dict(n=n, mol=mol, gset=gset)


def test_qm9_sized_dataset_is_coarsened_and_assembled_whole(qm9):
    """The whole dataset in one batched contraction (one wavefront per molecule) + pooling + subgraph assembly: every node is
    the own node of exactly one cluster subgraph, every molecule keeps >= 1 cluster and about half its nodes, the
    reference's row mask marks as many rows as there are nodes (utils.py:498-503)."""
    mol, gset, n = qm9["mol"], qm9["gset"], qm9["n"]
    N = int(mol["node_ptr"][-1])
    assert gset.n_graphs == n and int(gset.gs_core.sum()) == N and int(gset.gs_mask.sum()) == N
    per_graph = np.diff(gset.cluster_ptr)
    nodes = np.diff(mol["node_ptr"])
    assert per_graph.min() >= 1 and np.all(per_graph <= nodes)
    assert abs(per_graph.sum() / N - 0.5) < 0.05, "r = 0.5 (main.py:370-377 passes 1 - coarsening_ratio)"
    own = gset.gs_node[gset.gs_core]
    assert int(torch.unique(own).numel()) == N


def test_qm9_sampled_batches_match_the_literal_per_subgraph_loops(qm9):
    """Regress_graph_gs on 128-graph loader batches taken from the start, the middle and the end of the training half
    (run.py:513: batch_size 128), one block-diagonal pass each at hidden 512, against the oracle's literal restatement of
    network.py:189-204 (a conv stack per subgraph, x[mask], mean pool per graph, lt1): outputs <= 1e-4, gradients <= 1e-3."""
    import types

    from fitgnn_amd import network, train
    from oracle import gnn_oracle as gorc

    mol, gset, n = qm9["mol"], qm9["gset"], qm9["n"]
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=512, num_classes=1, dropout=0.0)
    torch.manual_seed(4)
    model = network.Regress_graph_gs(args).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    y_all = torch.from_numpy(mol["y"])
    e_all = gset.gs_edge_index
    half = n // 2
    for g0 in (0, (half // 2) // 128 * 128, (half - 128) // 128 * 128):
        b = train._cat_pieces([gset.batch(g0, g0 + 128, "gs")], "gs", types)   # the batch dict GraphTrainer feeds the model
        model.zero_grad()
        out = model(b, b["graph_of_masked"])
        tgt = y_all[g0:g0 + 128].long()[:, 0].view(-1, 1).float().cuda()
        loss = torch.nn.functional.l1_loss(out, tgt)
        loss.backward()
        # the same batch in the reference's layout: per graph, per cluster subgraph
        set_gs = []
        r_lo, r_hi = int(gset.gs_ptr[g0]), int(gset.gs_ptr[g0 + 128])
        keep = (e_all[0] >= r_lo) & (e_all[0] < r_hi)
        e = e_all[:, keep].cpu()
        for g in range(g0, g0 + 128):
            subs = []
            for c in range(int(gset.cluster_ptr[g]), int(gset.cluster_ptr[g + 1])):
                r0, r1 = int(gset.sub_ptr[c]), int(gset.sub_ptr[c + 1])
                k = (e[0] >= r0) & (e[0] < r1)
                subs.append(dict(x=gset.gs_x[r0:r1].cpu(), edge_index=e[:, k] - r0, mask=gset.gs_mask[r0:r1].cpu()))
            set_gs.append(subs)
        bt = torch.repeat_interleave(torch.arange(128), torch.from_numpy(np.diff(mol["node_ptr"][g0:g0 + 129])))
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = gorc.regress_graph_gs_forward(params, set_gs, bt)
        ref_loss = torch.nn.functional.l1_loss(ref, tgt.cpu())
        ref_loss.backward()
        assert rel(out.detach().cpu(), ref.detach()) < 1e-4, g0
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
        for k, p in model.named_parameters():
            assert rel(p.grad.cpu(), params[k].grad) < 1e-3, (g0, k)


def test_qm9_training_epoch_from_captured_steps(qm9):
    """One Gs training epoch over the 65 415 training graphs (utils.py:33: half of the dataset), 512 steps of 128 graphs
    replayed from hipGraphs, against the SAME epoch run eagerly from the same weights (dropout off, so that the two do not
    depend on how their dropout seeds are drawn): the reported epoch loss and the weights after the epoch agree; then a second
    captured epoch with dropout on: finite loss, weights keep moving (the reference never clears the gradients inside an epoch,
    run.py:257,291 -- reproduced -- so the loss need not fall from one epoch to the next at this step count)."""
    from fitgnn_amd import network, train

    gset, n = qm9["gset"], qm9["n"]
    graphs = list(range(n // 2))
    runs = {}
    for capture in (True, False):
        args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=512, num_classes=1, dropout=0.0)
        torch.manual_seed(5)
        model = network.Regress_graph_gs(args).cuda()
        tr = train.GraphTrainer(model, gset, graphs, kind="gs", batch_size=128, lr=0.001, capture=capture)
        assert len(tr.batches) == (n // 2 + 127) // 128
        loss = float(tr.step())
        runs[capture] = (loss, {k: v.detach().clone() for k, v in model.state_dict().items()})
    l_cap, w_cap = runs[True]
    l_eag, w_eag = runs[False]
    assert np.isfinite(l_cap) and abs(l_cap - l_eag) <= 1e-4 * abs(l_eag), (l_cap, l_eag)
    for k in w_cap:
        assert rel(w_cap[k], w_eag[k]) < 1e-3, k
    # dropout on (device-resident seeds advanced inside the captured steps)
    args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=512, num_classes=1)
    torch.manual_seed(5)
    model = network.Regress_graph_gs(args).cuda()
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = train.GraphTrainer(model, gset, graphs, kind="gs", batch_size=128, lr=0.001, capture=True)
    l1 = float(tr.step())
    l2 = float(tr.step())
    assert np.isfinite(l1) and np.isfinite(l2)
    assert any(not torch.equal(v, w0[k]) for k, v in model.state_dict().items())
