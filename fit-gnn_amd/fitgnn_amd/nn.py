"""Message-passing layers with torch_geometric.nn's call signatures, on the fitgnn HIP kernels.

FIT-GNN resolves its layer class by name -- `getattr(torch_geometric.nn, args.layer_name)`
(network.py:13,41,70,101,141,172) -- and calls `conv(x, edge_index)`.  This module is the namespace
that lookup is pointed at instead: same constructor arguments, same `forward(x, edge_index)`, same
parameter names (so `state_dict`s interchange: `lin.weight [out,in]`, `bias [out]` for GCNConv -- the
layout pinned by Baselines/SGGC/GCN/params/checkpoint-best-acc.pkl).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .csr import csr_for


class _OpConfigured:
    """Mixin: the ops.OpConfig a layer's kernels run under (`op_config`; the package default unless set on the instance,
    e.g. by network._Base.set_op_config)."""
    op_config = ops.DEFAULT


def glorot_(w):
    a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
    with torch.no_grad():
        w.uniform_(-a, a)


class GCNConv(_OpConfigured, nn.Module):
    """out = D^-1/2 (A+I) D^-1/2 (x W^T) + b   (PyG GCNConv defaults: add_self_loops, normalize, bias)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def graph(self, edge_index, num_nodes):
        return csr_for(edge_index, num_nodes, "gcn")

    def forward(self, x, edge_index):
        g = self.graph(edge_index, x.shape[0])
        h = ops.Linear.apply(x.float(), self.lin.weight, self.op_config)
        return ops.SpMM.apply(h, self.bias, g, self.op_config)

    def forward_elu_dropout(self, x, edge_index, p=0.5, training=False, mask=None, graph=None, link_in=None, link_out=None):
        """conv -> F.elu -> F.dropout (network.py:31-33) as one GEMM + one SpMM with fused epilogue.
        link_in / link_out (ops.EpilogueLink): x is the un-shared output of the previous fused layer / the output goes to
        exactly one next fused layer (sequential stacks only, see ops.EpilogueLink)."""
        g = graph if graph is not None else self.graph(edge_index, x.shape[0])
        cfg = self.op_config
        seed = ops.next_seed(cfg) if (training and p > 0 and mask is None) else 0
        return ops.FusedGCNLayer.apply(x, self.lin.weight, self.bias, g, float(p), bool(training), seed, mask, link_in, link_out, cfg)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}"


class Linear(_OpConfigured, nn.Linear):
    """torch.nn.Linear (same parameters, same state_dict keys) whose products run under ops' GEMM policy on the GPU:
    3 x bf16-split MFMA for x W^T and dy W, the split-K batched product for the weight gradient dy^T x (torch's own
    fp32 path takes 400-540 us per product on a 90 k-row batch, 2.5x longer)."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 2:
            y = ops.Linear.apply(x.float(), self.weight, self.op_config)
            if self.bias is None:
                return y
            return ops.BiasAdd.apply(y, self.bias) if (self.out_features <= 64 and y.shape[0] > 4096) else y + self.bias
        return super().forward(x)


class SAGEConv(_OpConfigured, nn.Module):
    """out = W_l mean_{j in N(i)} x_j + b_l + W_r x_i   (PyG SAGEConv defaults: aggr='mean', root_weight)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, edge_index):
        g = csr_for(edge_index, x.shape[0], "mean")
        x = x.float()
        if self.in_channels <= self.out_channels:  # aggregate in the narrower space; mean and Linear commute
            agg = ops.SpMM.apply(x, None, g, self.op_config)
            return self.lin_l(agg) + self.lin_r(x)
        h = ops.Linear.apply(x, self.lin_l.weight, self.op_config)
        return ops.SpMM.apply(h, None, g, self.op_config) + (self.lin_l.bias if self.lin_l.bias is not None else 0.0) + self.lin_r(x)


class GINConv(_OpConfigured, nn.Module):
    """out = nn((1 + eps) x_i + sum_{j in N(i)} x_j)   (PyG GINConv; FIT-GNN passes train_eps=True)."""

    def __init__(self, nn_module, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.empty(1))
        else:
            self.register_buffer("eps", torch.empty(1))
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.nn.modules():
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        with torch.no_grad():
            self.eps.fill_(self.initial_eps)

    def forward(self, x, edge_index):
        g = csr_for(edge_index, x.shape[0], "sum")
        x = x.float()
        return self.nn((1.0 + self.eps) * x + ops.SpMM.apply(x, None, g, self.op_config))


class GATConv(_OpConfigured, nn.Module):
    """PyG GATConv with the arguments FIT-GNN passes (network.py:13-17: `GATConv(in, out)` -> heads=1, concat,
    negative_slope=0.2, dropout=0, add_self_loops, bias).  Parameters: lin.weight [out,in], att_src [1,1,out],
    att_dst [1,1,out], bias [out] (PyG >= 2.4 naming)."""

    def __init__(self, in_channels, out_channels, heads=1, negative_slope=0.2, bias=True):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("FIT-GNN constructs GATConv with the default single head")
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        glorot_(self.att_src)
        glorot_(self.att_dst)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, edge_index):
        g = csr_for(edge_index, x.shape[0], "gat")
        cfg = self.op_config
        h = ops.Linear.apply(x.float(), self.lin.weight, cfg)
        return ops.GATAggregate.apply(h, self.att_src.view(-1), self.att_dst.view(-1), self.bias, g, self.negative_slope,
                                      False, 0.0, False, 0, None, cfg)

    def forward_elu_dropout(self, x, edge_index, p=0.5, training=False, mask=None, x_index=None, link_out=None):
        """conv -> F.elu -> F.dropout (network.py:31-33) with the activation in the aggregation kernel's epilogue.
        x_index (ops.RowIndex, optional): x is a de-duplicated feature table and node r of the graph is a copy of table row
        x_index.index[r]: the Linear and the score dots run on the table (ops.GATAggregate, ridx).
        link_out (ops.EpilogueLink): the output goes to exactly one consumer that may apply this layer's ELU' / dropout' itself."""
        n = x.shape[0] if x_index is None else int(x_index.index.numel())
        g = csr_for(edge_index, n, "gat")
        cfg = self.op_config
        h = ops.Linear.apply(x.float(), self.lin.weight, cfg)
        seed = ops.next_seed(cfg) if (training and p > 0 and mask is None) else 0
        return ops.GATAggregate.apply(h, self.att_src.view(-1), self.att_dst.view(-1), self.bias, g, self.negative_slope,
                                      True, float(p), bool(training), seed, mask, cfg, x_index, link_out)


class APPNP(_OpConfigured, nn.Module):
    """z <- (1-alpha) A_hat z + alpha z0, K times (Baselines/SGGC/APPNP/networks.py:11,23: K=10, alpha=0.1)."""

    def __init__(self, K, alpha, dropout=0.0):
        super().__init__()
        self.K, self.alpha, self.dropout = K, alpha, dropout

    def reset_parameters(self):
        pass

    def forward(self, x, edge_index, x_index=None):
        """x_index (ops.RowIndex, optional, an extension): x is a de-duplicated table and row r of the propagated signal is
        x[x_index.index[r]] (edge_index numbers the signal's rows)."""
        n = x.shape[0] if x_index is None else int(x_index.index.numel())
        g = csr_for(edge_index, n, "gcn")
        z0 = x.float()
        if z0.is_cuda and z0.shape[1] <= 64:   # class-wide signal: the narrow kernel, teleport term in its epilogue
            return ops.APPNPPropagate.apply(z0, g, int(self.K), float(self.alpha), self.op_config, x_index)
        if x_index is not None:
            z0 = z0.index_select(0, x_index.index.long())
        z = z0
        for _ in range(self.K):
            z = ops.SpMM.apply(z, None, g, self.op_config) * (1.0 - self.alpha) + self.alpha * z0
        return z


def global_mean_pool(x, batch, size=None, rows=None):
    """torch_geometric.nn.global_mean_pool (network.py:164,202).  rows (int64 index, optional, an extension): pool x[rows], batch[i]
    being the graph of rows[i] -- the *_gs models' x[mask] without the gathered copy.  On the GPU with a sorted `batch` (PyG's batch
    vectors are): one gather-and-sum launch per pool (ops.SegmentMeanPool); otherwise the scatter form."""
    size = int(batch.max().item()) + 1 if size is None else size
    if ops.pool_supported(x):
        pi = ops.pool_index(batch, size, rows, x.shape[0])
        if pi.sorted:
            return ops.SegmentMeanPool.apply(x, pi)
    if rows is not None:
        x = x.index_select(0, rows)
    out = torch.zeros((size, x.shape[1]), dtype=x.dtype, device=x.device).index_add_(0, batch, x)
    cnt = torch.zeros(size, dtype=x.dtype, device=x.device).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


def global_max_pool(x, batch, size=None, rows=None):
    """torch_geometric.nn.global_max_pool (network.py:93,131); `rows` as in global_mean_pool."""
    size = int(batch.max().item()) + 1 if size is None else size
    if ops.pool_supported(x):
        pi = ops.pool_index(batch, size, rows, x.shape[0])
        if pi.sorted:
            return ops.SegmentMaxPool.apply(x, pi)
    if rows is not None:
        x = x.index_select(0, rows)
    out = torch.full((size, x.shape[1]), float("-inf"), dtype=x.dtype, device=x.device)
    return out.scatter_reduce(0, batch.unsqueeze(1).expand_as(x), x, reduce="amax", include_self=True)
