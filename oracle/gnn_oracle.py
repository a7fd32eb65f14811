"""CPU oracle for the train half (TEST INFRASTRUCTURE ONLY; never imported by fit-gnn_amd/).

torch_geometric is a third-party dependency of the reference that is NOT vendored under /root/reference
and is not installable here (requirements.txt:2,11,12 unpinned), so this half of the
oracle restates PyG's documented operator semantics in plain torch CPU ops (index_add_ gather/scatter,
exactly the data flow of MessagePassing.propagate) and follows, as text:
    network.py:8-64      layer order conv -> F.elu -> F.dropout(p=0.5) -> lt1 -> log_softmax
    run.py:177-215       GD step: concatenate masked outputs of every batch, ONE loss, one backward
    run.py:341-344       NLLLoss, Adam(lr, weight_decay=5e-4)
PARITY UNPINNED against PyG numerics (no reference test or fixture pins them); what IS pinned by the
reference is the parameter layout `<conv>.lin.weight [out,in]`, `<conv>.bias [out]`
(Baselines/SGGC/GCN/params/checkpoint-best-acc.pkl, checked in tests/test_oracle_gnn.py).
"""
import torch
import torch.nn.functional as F


def gcn_norm(edge_index, num_nodes, dtype=torch.float32):
    """torch_geometric.nn.conv.gcn_conv.gcn_norm with add_self_loops=True, unweighted input:
    add_remaining_self_loops (existing self loops are replaced by exactly one of weight 1),
    deg = scatter_add(w, col), norm = deg^-1/2[row] * w * deg^-1/2[col], inf -> 0."""
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop = torch.arange(num_nodes, dtype=row.dtype)
    row = torch.cat([row[keep], loop])
    col = torch.cat([col[keep], loop])
    w = torch.ones(row.numel(), dtype=dtype)
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, col, w)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    return row, col, dinv[row] * w * dinv[col]


def propagate(row, col, w, h, num_nodes):
    """out[i] = sum over edges (j -> i) of w * h[j]; row = source j, col = target i (PyG flow)."""
    return torch.zeros((num_nodes, h.shape[1]), dtype=h.dtype).index_add_(0, col, h[row] * w.unsqueeze(1))


def gcn_conv(x, edge_index, weight, bias):
    """GCNConv.forward: x W^T (Linear, no bias) -> propagate with gcn_norm -> + bias."""
    n = x.shape[0]
    row, col, w = gcn_norm(edge_index, n, x.dtype)
    out = propagate(row, col, w, x @ weight.t(), n)
    return out if bias is None else out + bias


def sage_conv(x, edge_index, w_l, b_l, w_r):
    """SAGEConv(aggr='mean', root_weight=True): lin_l(mean_j x_j) + lin_r(x_i)."""
    n = x.shape[0]
    row, col = edge_index[0], edge_index[1]
    agg = torch.zeros_like(x).index_add_(0, col, x[row])
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, col, torch.ones(row.numel(), dtype=x.dtype)).clamp(min=1)
    agg = agg / cnt.unsqueeze(1)
    out = agg @ w_l.t() + x @ w_r.t()
    return out if b_l is None else out + b_l


def gin_aggregate(x, edge_index, eps):
    """GINConv before its MLP: (1 + eps) x_i + sum_j x_j."""
    row, col = edge_index[0], edge_index[1]
    return (1.0 + eps) * x + torch.zeros_like(x).index_add_(0, col, x[row])


def gat_conv(x, edge_index, weight, att_src, att_dst, bias, negative_slope=0.2):
    """GATConv.forward, heads=1: h = x W^T; e = leaky_relu(a_src.h_j + a_dst.h_i); softmax over the incoming
    edges of i (self loops: existing ones removed, one added per node); out_i = sum alpha_ij h_j + bias."""
    n = x.shape[0]
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop = torch.arange(n, dtype=row.dtype)
    row, col = torch.cat([row[keep], loop]), torch.cat([col[keep], loop])
    h = x @ weight.t()
    a_s, a_d = (h * att_src.view(1, -1)).sum(1), (h * att_dst.view(1, -1)).sum(1)
    e = F.leaky_relu(a_s[row] + a_d[col], negative_slope)
    m = torch.full((n,), float("-inf"), dtype=x.dtype).scatter_reduce(0, col, e, reduce="amax", include_self=True)
    p = torch.exp(e - m[col])
    z = torch.zeros(n, dtype=x.dtype).index_add_(0, col, p)
    alpha = p / z[col]
    out = torch.zeros((n, h.shape[1]), dtype=x.dtype).index_add_(0, col, h[row] * alpha.unsqueeze(1))
    return out if bias is None else out + bias


def appnp(x, edge_index, K, alpha):
    """APPNP.forward (Baselines/SGGC/APPNP/networks.py:11,23): z <- (1-alpha) A_hat z + alpha z0."""
    n = x.shape[0]
    row, col, w = gcn_norm(edge_index, n, x.dtype)
    z = x
    for _ in range(K):
        z = propagate(row, col, w, z, n) * (1 - alpha) + alpha * x
    return z


def classify_node_gat_fwd_bwd(sd, x, edge_index, y, num_layers=2, train_mask=None, dtype=torch.float32):
    """network.py:29-35 with GATConv layers (--layer_name GATConv) in eval mode: forward, NLL loss, gradients."""
    params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    h = x.to(dtype)
    for i in range(num_layers):
        h = F.elu(gat_conv(h, edge_index, params[f"conv.{i}.lin.weight"], params[f"conv.{i}.att_src"], params[f"conv.{i}.att_dst"],
                           params.get(f"conv.{i}.bias")))
    out = F.log_softmax(h @ params["lt1.weight"].t() + params["lt1.bias"], dim=1)
    sel = out if train_mask is None else out[train_mask]
    tgt = y if train_mask is None else y[train_mask]
    loss = F.nll_loss(sel, tgt.long())
    loss.backward()
    return out.detach(), loss.detach(), {k: v.grad for k, v in params.items()}


def appnp_net_fwd_bwd(sd, x, edge_index, y, K=10, alpha=0.1, train_mask=None, dtype=torch.float32):
    """Baselines/SGGC/APPNP/networks.py:17-27 in eval mode (dropout is the identity): lin1 -> ReLU -> lin2 -> APPNP -> log_softmax,
    NLL loss, gradients of lin1 / lin2.  sd keys: lin1.weight, lin1.bias, lin2.weight, lin2.bias."""
    params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    h = F.relu(x.to(dtype) @ params["lin1.weight"].t() + params["lin1.bias"])
    z = h @ params["lin2.weight"].t() + params["lin2.bias"]
    out = F.log_softmax(appnp(z, edge_index, K, alpha), dim=1)
    sel = out if train_mask is None else out[train_mask]
    tgt = y if train_mask is None else y[train_mask]
    loss = F.nll_loss(sel, tgt.long())
    loss.backward()
    return out.detach(), loss.detach(), {k: v.grad for k, v in params.items()}


def classify_node_forward(sd, x, edge_index, num_layers, masks=None, p=0.5):
    """network.py:29-35 with GCNConv layers; `masks` (list of {0,1} tensors) injects dropout patterns
    (training mode); None = eval mode."""
    for i in range(num_layers):
        x = gcn_conv(x, edge_index, sd[f"conv.{i}.lin.weight"], sd.get(f"conv.{i}.bias"))
        x = F.elu(x)
        if masks is not None:
            x = x * masks[i].to(x.dtype) / (1.0 - p)
    x = x @ sd["lt1.weight"].t() + sd["lt1.bias"]
    return F.log_softmax(x, dim=1)


def classify_node_fwd_bwd(sd, x, edge_index, y, num_layers=2, train_mask=None, masks=None, dtype=torch.float32, loss_scale=None):
    """Forward, NLL loss (run.py:200,341) and gradients of every parameter.  loss_scale: the loss is loss_scale x the SUM
    over the train nodes instead of their mean (a slice of a larger union under the union's 1 / count, run.py:184-204)."""
    params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    out = classify_node_forward(params, x.to(dtype), edge_index, num_layers, masks=masks)
    sel = out if train_mask is None else out[train_mask]
    tgt = y if train_mask is None else y[train_mask]
    loss = F.nll_loss(sel, tgt.long()) if loss_scale is None else F.nll_loss(sel, tgt.long(), reduction="sum") * loss_scale
    loss.backward()
    return out.detach(), loss.detach(), {k: v.grad for k, v in params.items()}


def gd_train_step(sd, batches, num_layers=2, lr=0.01, weight_decay=5e-4, adam_state=None, masks=None):
    """One epoch of node_train_Gs_GD (run.py:177-215): forward every batch that has a train node,
    concatenate masked outputs, ONE nll loss, one backward, one Adam step.  batches: list of dicts with
    x, edge_index, y, train_mask.  Returns (loss, new_state_dict, adam_state)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    outs, labels = [], []
    for bi, b in enumerate(batches):
        if not bool(b["train_mask"].any()):
            continue
        out = classify_node_forward(params, b["x"], b["edge_index"], num_layers, masks=None if masks is None else masks[bi])
        outs.append(out[b["train_mask"]])
        labels.append(b["y"][b["train_mask"]])
    loss = F.nll_loss(torch.cat(outs), torch.cat(labels).long())
    loss.backward()
    plist = list(params.values())
    opt = torch.optim.Adam(plist, lr=lr, weight_decay=weight_decay)
    if adam_state is not None:
        opt.load_state_dict(adam_state)
    opt.step()
    return loss.detach(), {k: v.detach() for k, v in params.items()}, opt.state_dict()


def mb_train_epoch(sd, batches, num_layers=2, lr=0.01, weight_decay=5e-4, adam_state=None, reduction="mean"):
    """One epoch of node_train_Gs_MB (run.py:217-252): zero_grad ONCE (:222), then per batch with a train node:
    forward, loss, backward, optimizer.step() -- gradients accumulate across the epoch's batches.  Dropout off
    (the caller compares in p=0 mode).  Returns (reported loss, new_state_dict, adam_state)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, weight_decay=weight_decay)
    if adam_state is not None:
        opt.load_state_dict(adam_state)
    opt.zero_grad()
    total, n_out = 0.0, 0
    for b in batches:
        if not bool(b["train_mask"].any()):
            continue
        out = classify_node_forward(params, b["x"], b["edge_index"], num_layers, masks=None, p=0.0)
        loss = F.nll_loss(out[b["train_mask"]], b["y"][b["train_mask"]].long(), reduction=reduction)
        loss.backward()
        opt.step()
        total += float(loss)
        n_out += int(b["train_mask"].sum())
    rep = total / len(batches) if reduction == "mean" else total / max(n_out, 1)
    return rep, {k: v.detach() for k, v in params.items()}, opt.state_dict()


def conv_stack(sd, x, edge_index, num_layers):
    """for l < L: conv -> elu (-> dropout, off here): the body shared by every model of network.py."""
    for i in range(num_layers):
        x = F.elu(gcn_conv(x, edge_index, sd[f"conv.{i}.lin.weight"], sd.get(f"conv.{i}.bias")))
    return x


def mean_pool(x, batch, size):
    out = torch.zeros((size, x.shape[1]), dtype=x.dtype).index_add_(0, batch, x)
    cnt = torch.zeros(size, dtype=x.dtype).index_add_(0, batch, torch.ones(batch.numel(), dtype=x.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


def regress_graph_gs_forward(sd, set_gs, batch_tensor, num_layers=2):
    """Regress_graph_gs.forward (network.py:189-204), literally: every subgraph of every graph runs the conv stack on
    its OWN (its own gcn_norm), x[mask] rows are concatenated in loop order, mean-pooled per graph, then lt1.
    set_gs: list (graphs) of lists (subgraphs) of dicts x, edge_index, mask."""
    rows = [conv_stack(sd, g["x"].float(), g["edge_index"], num_layers)[g["mask"]] for gs in set_gs for g in gs]
    x = mean_pool(torch.cat(rows, 0), batch_tensor.long(), len(set_gs))
    return x @ sd["lt1.weight"].t() + sd["lt1.bias"]


def regress_graph_gc_forward(sd, gc_x, gc_edge_index, gc_batch, n_graphs, num_layers=2):
    """Regress_graph_gc.forward (network.py:156-164)."""
    x = mean_pool(conv_stack(sd, gc_x, gc_edge_index, num_layers), gc_batch.long(), n_graphs)
    return x @ sd["lt1.weight"].t() + sd["lt1.bias"]


def graph_train_epoch(sd, batches, forward, prop=0, lr=0.01, weight_decay=5e-4, adam_state=None):
    """graph_train_Gs / graph_train_Gc (run.py:254-269, :288-304), multi_prop regression: zero_grad ONCE per epoch,
    then per batch: y.type(torch.long) (truncation!), L1Loss(out, y[:, prop].view(-1, 1)), backward, step.
    batches: list of (inputs..., y); forward(params, *inputs) -> [B, 1].  Returns (mean batch loss, sd, adam)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, weight_decay=weight_decay)
    if adam_state is not None:
        opt.load_state_dict(adam_state)
    opt.zero_grad()
    total = 0.0
    for *inputs, y in batches:
        out = forward(params, *inputs)
        loss = F.l1_loss(out, y.long()[:, prop].view(-1, 1).to(out.dtype))
        loss.backward()
        opt.step()
        total += float(loss)
    return total / len(batches), {k: v.detach() for k, v in params.items()}, opt.state_dict()
