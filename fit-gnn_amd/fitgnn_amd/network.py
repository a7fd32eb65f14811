"""The six model classes of FIT-GNN's network.py, same surface, on fitgnn_amd.nn layers.

Kept from the reference (network.py:8-204): constructor takes the argparse Namespace (num_layers1,
layer_name, num_features, hidden, num_classes); attributes `.conv` (ModuleList) and `.lt1` (Linear);
`state_dict` keys `conv.{i}.*`, `lt1.*`; forward signatures and output activations:
    Classify_node(x, edge_index)      -> log_softmax            (:29-35)
    Regress_node(x, edge_index)       -> raw [N,1]               (:58-64)
    Classify_graph_gc(gc)             -> softmax(max-pool)       (:87-95)
    Regress_graph_gc(gc)              -> lt1(mean-pool)          (:158-166)
    Classify_graph_gs(set_gs, batch)  -> softmax(max-pool of masked subgraph rows)   (:118-135)
    Regress_graph_gs(set_gs, batch)   -> lt1(mean-pool of masked subgraph rows)      (:189-204)
Each layer is conv -> ELU -> dropout(p=0.5) (network.py:31-33); GCN layers run that as one GEMM plus one
SpMM with a fused epilogue.  The *_gs classes evaluate all subgraphs of a batch as ONE block-diagonal
pass instead of the reference's Python double loop (identical arithmetic: subgraphs share no edges).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import nn as fnn
from . import ops


def _make_convs(args):
    cls = getattr(fnn, args.layer_name, None)
    if cls is None:
        raise AttributeError(f"fitgnn_amd.nn has no layer '{args.layer_name}'")
    convs = nn.ModuleList()
    dims = [args.num_features] + [args.hidden] * args.num_layers1
    for i in range(args.num_layers1):
        if args.layer_name == "GINConv":  # network.py:19-21: two-layer ReLU MLP, train_eps=True
            mlp = nn.Sequential(fnn.Linear(dims[i], args.hidden), nn.ReLU(), fnn.Linear(args.hidden, args.hidden), nn.ReLU())
            convs.append(cls(mlp, train_eps=True))
        else:
            convs.append(cls(dims[i], dims[i + 1]))
    return convs


class _Base(nn.Module):
    out_dim_from_classes = True

    def __init__(self, args):
        super().__init__()
        self.num_layers = args.num_layers1
        self.conv = _make_convs(args)
        self.lt1 = nn.Linear(args.hidden, args.num_classes if self.out_dim_from_classes else 1)
        self.dropout_p = float(getattr(args, "dropout", 0.5))  # F.dropout's default p, which is what the reference uses
        self._inject_masks = None  # tests: list of uint8 masks, one per layer
        self.op_config = ops.DEFAULT

    def set_op_config(self, cfg):
        """Run this model's kernels under `cfg` (ops.OpConfig): GEMM policy, kernel A/B switches, profiling hooks, seed
        bank.  Per model: two models in one process keep their own."""
        self.op_config = cfg
        for m in self.modules():
            if isinstance(m, fnn._OpConfigured):
                m.op_config = cfg
        return self

    def reset_parameters(self):
        for m in self.conv:
            m.reset_parameters()
        self.lt1.reset_parameters()

    def _first_layer_dedup(self, x_table, edge_index, x_index, link_out=None, gat_link=None):
        """Layer 0 on a de-duplicated feature table (x_index: ops.RowIndex mapping union rows to table rows).
        gat_link: the link an attention layer records for an aggregate-first attention layer right behind it (FusedGATLastLayerRows)."""
        conv = self.conv[0]
        if isinstance(conv, fnn.GATConv) and x_table.is_cuda and link_out is None:
            mask = self._inject_masks[0] if self._inject_masks is not None else None
            return conv.forward_elu_dropout(x_table, edge_index, p=self.dropout_p, training=self.training, mask=mask, x_index=x_index,
                                            link_out=gat_link)
        if not (isinstance(conv, fnn.GCNConv) and x_table.is_cuda):
            return None
        mask = self._inject_masks[0] if self._inject_masks is not None else None
        g = conv.graph(edge_index, int(x_index.index.numel()))
        cfg = self.op_config
        seed = ops.next_seed(cfg) if (self.training and self.dropout_p > 0 and mask is None) else 0
        return ops.FusedGCNLayerDedup.apply(x_table.float(), conv.lin.weight, conv.bias, g, x_index, float(self.dropout_p),
                                            bool(self.training), seed, mask, link_out, cfg)

    def prepare_static(self, x, edge_index, pooled_rows=None):
        """Form ahead of time what the layers keep per (graph, input) -- A_hat x of a narrow static input (ops.aggregated_input), the
        compact positions of the pooled rows (ops.FusedGCNLayerRows' backward) -- e.g. before the steps over a
        set of static batches are captured in hipGraphs: made lazily inside a step they would be captured and replayed with it."""
        conv = self.conv[0] if self.num_layers > 0 else None
        if (isinstance(conv, fnn.GCNConv) and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
                and ops.narrow_input_supported(x, conv.lin.weight, self.op_config)):
            ops.aggregated_input(conv.graph(edge_index, x.shape[0]), x, self.op_config)
        last = self.conv[self.num_layers - 1] if self.num_layers > 0 else None
        if (pooled_rows is not None and torch.is_tensor(x) and x.is_cuda and isinstance(last, fnn.GCNConv) and self.op_config.pooled_rows_last_layer
                and pooled_rows.dtype == torch.int64 and pooled_rows.numel() > 0):
            ops._compact_positions(last.graph(edge_index, x.shape[0]), pooled_rows)

    def embed(self, x, edge_index, x_index=None, first=0, link=None, last=None, return_link=False, tail_link=None):
        """conv -> ELU -> dropout, layers first .. last - 1 (default: all; network.py:29-33).  link: the EpilogueLink recorded by the
        layer that produced x (consecutive fused GCN layers are linked: see ops.EpilogueLink; the stack is strictly sequential).
        return_link: also return the link the last evaluated layer recorded for its ONE consumer (None when it recorded none).
        tail_link: the link the LAST evaluated layer records when it is an attention layer (its consumer: FusedGATLastLayerRows)."""
        x = x.float()
        for i in range(first, self.num_layers if last is None else last):
            conv = self.conv[i]
            if isinstance(conv, fnn.GCNConv) and x.is_cuda:
                mask = self._inject_masks[i] if self._inject_masks is not None else None
                nxt = ops.EpilogueLink() if i + 1 < self.num_layers else None   # the last output goes to lt1: plain gradient
                if i == 0 and link is None and ops.narrow_input_supported(x, conv.lin.weight, self.op_config):
                    # a narrow static input (QM9's 11 atom features): aggregate first, A_hat x formed once per (graph, input)
                    cfg = self.op_config
                    ax = ops.aggregated_input(conv.graph(edge_index, x.shape[0]), x, cfg)
                    seed = ops.next_seed(cfg) if (self.training and self.dropout_p > 0 and mask is None) else 0
                    x = ops.FusedGCNLayerAggregatedInput.apply(ax, conv.lin.weight, conv.bias, float(self.dropout_p), bool(self.training),
                                                               seed, mask, nxt, cfg)
                else:
                    x = conv.forward_elu_dropout(x, edge_index, p=self.dropout_p, training=self.training, mask=mask, link_in=link,
                                                 link_out=nxt)
                link = nxt
            elif isinstance(conv, fnn.GATConv) and x.is_cuda:
                mask = self._inject_masks[i] if self._inject_masks is not None else None
                is_tail = i + 1 == (self.num_layers if last is None else last)
                x = conv.forward_elu_dropout(x, edge_index, p=self.dropout_p, training=self.training, mask=mask,
                                             link_out=tail_link if is_tail else None)
                link = None
            else:
                x = conv(x, edge_index)
                x = F.elu(x)
                x = F.dropout(x, p=self.dropout_p, training=self.training)
                link = None
        return (x, link) if return_link else x

    def embed_pooled_rows(self, x, edge_index, rows):
        """embed() for a consumer that reads only `rows` of the result (the *_graph_gs models' pool over x[mask], network.py:129-131,
        :200-202): the last GCN layer aggregate-first on those rows, output compact [len(rows), hidden] (ops.FusedGCNLayerRows).
        None where that does not apply (the caller then embeds every row)."""
        L, cfg = self.num_layers, self.op_config
        last = self.conv[L - 1] if L > 0 else None
        if not (cfg.pooled_rows_last_layer and x.is_cuda and isinstance(last, fnn.GCNConv) and rows.dtype == torch.int64 and rows.numel() > 0
                and last.lin.weight.shape[0] % 4 == 0 and all(isinstance(c, (fnn.GCNConv, fnn.GATConv)) for c in self.conv)):
            return None
        if L > 1:
            h, link = self.embed(x, edge_index, last=L - 1, return_link=True)
        else:
            h, link = x.float(), None
        if h.shape[1] % 4 != 0 and link is not None:
            link = None
        mask = self._inject_masks[L - 1] if self._inject_masks is not None else None
        g = last.graph(edge_index, h.shape[0])
        seed = ops.next_seed(cfg) if (self.training and self.dropout_p > 0 and mask is None) else 0
        return ops.FusedGCNLayerRows.apply(h, last.lin.weight, last.bias, g, float(self.dropout_p), bool(self.training), seed, mask, rows, cfg,
                                           link)

    def embed_and_head(self, x, edge_index, x_index=None, out_rows=None, loss_rows=None, compact_logits=False, forward_rows_only=False):
        """embed() followed by lt1; on the GPU the last GCN layer and the head form one autograd node.
        x_index (ops.RowIndex, optional): x is a de-duplicated table and union row r is table row x_index.index[r].
        out_rows (csr.RowSubset, optional): return the head's output on those rows only.  The last layer is then
        evaluated as (A_hat[rows] h) W^T -- aggregation first, on the kept rows, so that its GEMM, activation and the
        head run on len(rows) rows instead of all of them (same values on those rows: A (h W^T) = (A h) W^T).
        loss_rows (int64 index tensor, optional): the caller's promise that only these rows of the result reach its loss
        (run.py:193-204 keeps out[mask]), i.e. that the gradient it sends back is zero elsewhere; the head's weight and bias
        gradients are then reduced over those rows alone, and the head itself is evaluated on those rows alone (the result is
        zero on the others; the GCN layers still compute every row).
        compact_logits (with loss_rows): the caller accepts the result as [len(loss_rows), C] (the logits of those rows, in their
        order) when the last layer runs on the loss rows -- check the returned shape: other paths return all rows.
        forward_rows_only (with loss_rows, sorted ascending): the last layer's forward aggregation runs on the loss rows alone
        (A_hat[rows, :] h: the rows nobody reads are not computed at all -- the pruned step); ignored where the aggregate-first
        last layer does not apply."""
        L = self.num_layers
        if out_rows is not None:
            return self._embed_and_head_rows(x, edge_index, x_index, out_rows)
        first = 0
        last = self.conv[L - 1] if L > 0 else None
        fused_tail = (L > 0 and x.is_cuda and isinstance(last, fnn.GCNConv) and
                      ops.head_fusable(self.lt1.in_features, self.lt1.out_features))
        # the conv stack is strictly sequential (network.py:29-33): consecutive fused GCN layers share an EpilogueLink, so
        # that the backward GEMM dH @ W of layer i+1 applies layer i's ELU'/dropout' in its epilogue
        link = ops.EpilogueLink() if (L > 1 and x.is_cuda and isinstance(self.conv[1], fnn.GCNConv)) else None
        # an attention layer right below an aggregate-first attention layer hands its ELU' / dropout' to that layer's adjoint aggregation
        glink = None
        if (not fused_tail and L > 1 and x.is_cuda and isinstance(last, fnn.GATConv) and isinstance(self.conv[L - 2], fnn.GATConv)
                and loss_rows is not None and self.op_config.last_layer_on_loss_rows and loss_rows.numel() > 0):
            glink = ops.EpilogueLink()
        if x_index is not None:
            h = self._first_layer_dedup(x, edge_index, x_index, link_out=link, gat_link=glink if L == 2 else None) if L > 1 else None
            if h is None:
                x = x.index_select(0, x_index.index.long())  # materialise the union rows
                link = None
            else:
                x, first = h, 1
        else:
            link = None
        if not fused_tail:
            cfg = self.op_config
            if (L > 0 and x.is_cuda and isinstance(last, fnn.GATConv) and loss_rows is not None and cfg.last_layer_on_loss_rows
                    and loss_rows.numel() > 0 and last.lin.weight.shape[0] % 4 == 0 and last.lin.weight.shape[1] % 4 == 0
                    and ops.head_rows_supported(x.new_empty((1, last.lin.weight.shape[0])), self.lt1.weight)
                    and ops.head_fusable(self.lt1.in_features, self.lt1.out_features)
                    and (first == L - 1 or isinstance(self.conv[L - 2], (fnn.GATConv, fnn.GCNConv)))):
                # the last attention layer aggregate-first: its dense part on the loss rows only (ops.FusedGATLastLayerRows)
                x = self.embed(x, edge_index, first=first, link=link, last=L - 1, tail_link=glink)
                mask = self._inject_masks[L - 1] if self._inject_masks is not None else None
                g = fnn.csr_for(edge_index, x.shape[0], "gat")
                seed = ops.next_seed(cfg) if (self.training and self.dropout_p > 0 and mask is None) else 0
                used = glink if (glink is not None and glink.epi != 0) else None   # (recorded by the layer below: it is an attention layer with act)
                return ops.FusedGATLastLayerRows.apply(x, last.lin.weight, last.att_src.view(-1), last.att_dst.view(-1), last.bias,
                                                       self.lt1.weight, self.lt1.bias, g, last.negative_slope, float(self.dropout_p),
                                                       bool(self.training), seed, mask, loss_rows, cfg, bool(compact_logits), used)
            return self.head(self.embed(x, edge_index, first=first, link=link))
        x = x.float()
        for i in range(first, L - 1):
            conv = self.conv[i]
            if isinstance(conv, fnn.GCNConv):
                mask = self._inject_masks[i] if self._inject_masks is not None else None
                nxt = ops.EpilogueLink()
                x = conv.forward_elu_dropout(x, edge_index, p=self.dropout_p, training=self.training, mask=mask, link_in=link,
                                             link_out=nxt)
                link = nxt
            else:
                x = F.dropout(F.elu(conv(x, edge_index)), p=self.dropout_p, training=self.training)
                link = None
        mask = self._inject_masks[L - 1] if self._inject_masks is not None else None
        g = last.graph(edge_index, x.shape[0])
        cfg = self.op_config
        seed = ops.next_seed(cfg) if (self.training and self.dropout_p > 0 and mask is None) else 0
        if (loss_rows is not None and cfg.last_layer_on_loss_rows and loss_rows.numel() > 0 and x.shape[1] % 4 == 0
                and last.lin.weight.shape[0] % 4 == 0 and ops.head_rows_supported(x.new_empty((1, last.lin.weight.shape[0])), self.lt1.weight)):
            # aggregate first, then the dense part on the loss rows only; the previous layer's epilogue backward rides on the
            # backward SpMM's store (link)
            fwd_sub = None
            if forward_rows_only:   # A_hat[rows, :] and its tiles, built once per (graph, loss rows)
                cache = getattr(g, "_rows_fwd", None)
                if not ops._same_index(cache, loss_rows):
                    from .csr import RowSubset
                    cache = (loss_rows, loss_rows._version, RowSubset(g, loss_rows))
                    g._rows_fwd = cache
                fwd_sub = cache[2]
            return ops.FusedGCNLastLayerRows.apply(x, last.lin.weight, last.bias, self.lt1.weight, self.lt1.bias, g,
                                                   float(self.dropout_p), bool(self.training), seed, mask, loss_rows, cfg, link,
                                                   bool(compact_logits), fwd_sub)
        return ops.FusedGCNLayerHead.apply(x, last.lin.weight, last.bias, self.lt1.weight, self.lt1.bias, g,
                                           float(self.dropout_p), bool(self.training), seed, mask, link, cfg, loss_rows)

    def _embed_and_head_rows(self, x, edge_index, x_index, sub):
        L = self.num_layers
        last = self.conv[L - 1]
        if not (L > 0 and isinstance(last, fnn.GCNConv)):
            raise NotImplementedError("out_rows needs a GCNConv last layer")
        first = 0
        if x_index is not None:
            h = self._first_layer_dedup(x, edge_index, x_index) if L > 1 else None
            if h is None:
                x = x.index_select(0, x_index.index.long())
            else:
                x, first = h, 1
        x = x.float()
        for i in range(first, L - 1):
            conv = self.conv[i]
            mask = self._inject_masks[i] if self._inject_masks is not None else None
            x = conv.forward_elu_dropout(x, edge_index, p=self.dropout_p, training=self.training, mask=mask) \
                if isinstance(conv, fnn.GCNConv) else F.dropout(F.elu(conv(x, edge_index)), p=self.dropout_p, training=self.training)
        agg = ops.SpMMRows.apply(x, sub, self.op_config)                   # [m, H_in]
        z = ops.Linear.apply(agg, last.lin.weight, self.op_config)
        if last.bias is not None:
            z = z + last.bias
        mask = self._inject_masks[L - 1] if self._inject_masks is not None else None
        z = F.elu(z)
        if mask is not None and self.training:
            z = z * mask.index_select(0, sub.rows).to(z.dtype) / (1.0 - self.dropout_p)
        else:
            z = F.dropout(z, p=self.dropout_p, training=self.training)
        return self.head(z)

    def head(self, x):
        """lt1 (network.py:34): same parameters as nn.Linear, evaluated as mm + broadcast add."""
        if x.is_cuda:
            return ops.SmallLinear.apply(x, self.lt1.weight, self.lt1.bias, self.op_config)
        return self.lt1(x)


class Classify_node(_Base):
    def forward(self, x, edge_index, x_index=None):
        return F.log_softmax(self.embed_and_head(x, edge_index, x_index), dim=1)


class Regress_node(_Base):
    out_dim_from_classes = False

    def forward(self, x, edge_index, x_index=None):
        return self.embed_and_head(x, edge_index, x_index)


class APPNPNet(nn.Module):
    """The APPNP model north_star names beside GCN / GAT: Baselines/SGGC/APPNP/networks.py:7-27 (`Net`), same attributes
    (`lin1`, `lin2`, `prop1`) and forward: dropout -> lin1 -> ReLU -> dropout -> lin2 -> APPNP(K, alpha) -> log_softmax.
    args: num_features, hidden, num_classes, K (default 10), alpha (default 0.1) (networks.py:9-11).
    x_index (ops.RowIndex, optional): x is a de-duplicated feature table and union row r is table row x_index.index[r] -- the MLP
    is per node, so it runs on the table and its class-wide output is gathered to the union rows before the propagation (in
    training mode the copies of a node then share its dropout draw)."""

    def __init__(self, args):
        super().__init__()
        self.lin1 = fnn.Linear(args.num_features, args.hidden)
        self.lin2 = fnn.Linear(args.hidden, args.num_classes)
        self.prop1 = fnn.APPNP(int(getattr(args, "K", 10)), float(getattr(args, "alpha", 0.1)))
        self.dropout_p = float(getattr(args, "dropout", 0.5))
        self.op_config = ops.DEFAULT

    def set_op_config(self, cfg):
        self.op_config = cfg
        for m in self.modules():
            if isinstance(m, fnn._OpConfigured):
                m.op_config = cfg
        return self

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def logits(self, x, edge_index, x_index=None):
        """forward() without its log_softmax: a trainer whose loss reads a few rows (run.py:193-204 keeps out[mask]) takes the
        softmax and the NLL on those rows only (ops.SoftmaxNLL) instead of normalising every union row twice."""
        x = F.dropout(x.float(), p=self.dropout_p, training=self.training)
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_p, training=self.training)
        x = self.lin2(x)
        return self.prop1(x, edge_index, x_index=x_index)   # (the table's rows are gathered into the propagation's own layout there)

    def forward(self, x, edge_index, x_index=None):
        return F.log_softmax(self.logits(x, edge_index, x_index), dim=1)


class Classify_graph_gc(_Base):
    def forward(self, gc):
        x = self.embed(gc.x, gc.edge_index)
        return F.softmax(self.head(fnn.global_max_pool(x, gc.batch, getattr(gc, "num_graphs", None))), dim=1)


def _mean_pool_head(model, x, batch, size, rows=None):
    """lt1(global_mean_pool(x[rows])): one launch each way on the GPU (ops.MeanPoolHead) when the batch vector is sorted and the
    shapes allow it, else the pool followed by the head."""
    if (model.op_config.fused_pool_head and size is not None and ops.pool_head_supported(x, model.lt1.weight)
            and batch.dtype == torch.int64):
        pi = ops.pool_index(batch, size, rows, x.shape[0])
        if pi.sorted:
            return ops.MeanPoolHead.apply(x, pi, model.lt1.weight, model.lt1.bias, model.op_config)
    if rows is not None:
        return model.head(fnn.global_mean_pool(x, batch, size, rows=rows))
    return model.head(fnn.global_mean_pool(x, batch, size))


class Regress_graph_gc(_Base):
    out_dim_from_classes = False

    def forward(self, gc):
        x = self.embed(gc.x, gc.edge_index)
        return _mean_pool_head(self, x, gc.batch.to(torch.int64), getattr(gc, "num_graphs", None))


def _merge_subgraphs(set_gs, device):
    """Block-diagonal union of every subgraph of every graph in the batch (reference: nested Python loops,
    network.py:120-130); returns x, edge_index, row mask in the reference's concatenation order."""
    xs, eis, masks, off = [], [], [], 0
    for gs in set_gs:
        for g in gs:
            xs.append(g.x.to(device).float())
            eis.append(g.edge_index.to(device) + off)
            masks.append(g.mask.to(device))
            off += g.x.shape[0]
    return torch.cat(xs, 0), torch.cat(eis, 1), torch.cat(masks, 0)


def _gs_inputs(set_gs, batch_tensor):
    """Reference form (list of per-graph lists of subgraph Data, network.py:120-130) or the pre-merged union a
    fitgnn_amd.graph_data.GraphSet batch carries (dict with x, edge_index, mask)."""
    if isinstance(set_gs, dict):
        return set_gs["x"], set_gs["edge_index"], set_gs.get("mask_idx", set_gs["mask"]), set_gs.get("n_graphs")
    return _merge_subgraphs(set_gs, batch_tensor.device) + (None,)


def _pooled_rows_ok(set_gs, mask):
    """The pooled rows come as a precomputed int64 index (GraphSet batches) and the batch does not opt out."""
    return isinstance(set_gs, dict) and mask is not None and mask.dtype == torch.int64 and not set_gs.get("_no_pooled_rows", False)


def _take(x, mask):
    """x[mask] for a bool mask, or index_select for a precomputed index (no host sync: usable under graph capture)."""
    return x.index_select(0, mask) if mask.dtype == torch.int64 else x[mask]


def _pool_rows(pool, x, mask, batch_tensor, size):
    """pool(x[mask]) per graph (network.py:129-131, :200-202); a precomputed row index (GraphSet batches) is folded into the pool."""
    if mask.dtype == torch.int64 and x.is_cuda:
        return pool(x, batch_tensor.to(torch.int64), size, rows=mask)
    return pool(_take(x, mask), batch_tensor.to(torch.int64), size)


class Classify_graph_gs(_Base):
    def forward(self, set_gs, batch_tensor):
        x, ei, mask, size = _gs_inputs(set_gs, batch_tensor)
        hc = self.embed_pooled_rows(x, ei, mask) if _pooled_rows_ok(set_gs, mask) else None
        if hc is not None:   # the last layer on the pooled rows only: the pool runs over the compact rows
            x = self.head(fnn.global_max_pool(hc, batch_tensor.to(torch.int64), size))
        else:
            x = self.head(_pool_rows(fnn.global_max_pool, self.embed(x, ei), mask, batch_tensor, size))
        return F.softmax(x, dim=0 if x.dim() == 1 else 1)


class Regress_graph_gs(_Base):
    out_dim_from_classes = False

    def forward(self, set_gs, batch_tensor):
        x, ei, mask, size = _gs_inputs(set_gs, batch_tensor)
        hc = self.embed_pooled_rows(x, ei, mask) if _pooled_rows_ok(set_gs, mask) else None
        if hc is not None:   # the last layer on the pooled rows only: pool and head over the compact rows
            return _mean_pool_head(self, hc, batch_tensor.to(torch.int64), size)
        h = self.embed(x, ei)
        if mask.dtype == torch.int64 and h.is_cuda:   # a precomputed row index (GraphSet batches): pool and head in one launch
            return _mean_pool_head(self, h, batch_tensor.to(torch.int64), size, rows=mask)
        return self.head(_pool_rows(fnn.global_mean_pool, h, mask, batch_tensor, size))
