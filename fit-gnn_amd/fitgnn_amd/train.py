"""Training-step functions of FIT-GNN's node-level path, on device-resident static batches.

Restates run.py:177-215 (`node_train_Gs_GD`): forward every batch that holds a train node, concatenate
the masked outputs, ONE loss over all of them, one backward, one optimiser step per epoch -- executed as
a single pass over the block-diagonal union (fitgnn_amd.data.SubgraphBatch).  Loss / optimiser as
run.py:341-344: NLLLoss(reduction=args.loss_reduction), Adam(lr, weight_decay=5e-4).
Data parallel (new functionality): subgraphs are sharded over ranks, every rank computes
sum-loss gradients scaled by 1/global_train_count, one flat all-reduce (RCCL) per step, replicated Adam.
"""
import torch
import torch.nn.functional as F


def _pad4(n):
    return (n + 3) // 4 * 4


class _accumulate_stream_guard:
    """Context for the warm-up steps that precede a hipGraph capture: tells whether autograd found a parameter's AccumulateGrad node bound
    to ANOTHER stream than the backward ran on (`.mismatch`).  Such a node outlived an earlier iteration -- a `loss` tensor of an earlier
    training phase that is still referenced keeps its graph's nodes alive -- and executes on the stream it was created on, usually the
    default stream: inside a capture that forks the default stream into the graph without a join, and ending the capture then takes the
    process down (found through main.py --exp_setup Gc_train_2_Gs_train, whose Gc phase trains the same parameters eagerly first).
    Autograd reports the condition as a warn-once UserWarning; the guard makes it warn always for its duration and records it, other
    warnings are passed on.  A trainer that sees `.mismatch` does not capture: it runs its steps eagerly."""
    KEY = "AccumulateGrad node's stream"

    def __enter__(self):
        import warnings
        self.mismatch = False
        self._prev = torch.is_warn_always_enabled()
        torch.set_warn_always(True)
        self._cw = warnings.catch_warnings(record=True)
        self._log = self._cw.__enter__()
        warnings.simplefilter("always")
        return self

    def __exit__(self, *exc):
        import warnings
        log = list(self._log)
        self._cw.__exit__(*exc)
        torch.set_warn_always(self._prev)
        for w in log:
            if self.KEY in str(w.message):
                self.mismatch = True
            else:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        return False


class FlatGrads:
    """All parameter gradients as views into one contiguous buffer -> one all-reduce per step.  Every parameter starts
    on a 16-byte boundary (sizes padded to multiples of four floats; the padding stays zero).  `buf` = the `n` gradient
    floats followed by a 4-float tail whose first slot carries the rank's loss share through the same all-reduce
    ("+ scalars for loss / count packed in")."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += _pad4(p.numel())
        self.n = n
        self.buf = torch.zeros(n + 4, dtype=torch.float32, device=self.params[0].device)
        self.grads, self.tail = self.buf[:n], self.buf[n:]
        for p, off in zip(self.params, self.offsets):
            p.grad = self.buf[off:off + p.numel()].view_as(p)
        self.fresh, self._fresh_views = None, None

    def zero(self):
        self.buf.zero_()

    def enable_fresh(self):
        """A second buffer of the same layout for the gradients of ONE backward pass (ops.OpConfig.grad_sink): backward nodes that
        know it store a weight gradient straight into its slice -- no `grad += new` launch per tensor -- and the optimiser kernel
        adds the buffer to the accumulated gradients and clears it (FlatAdam.step -> fitgnn_adam_step_acc_f32)."""
        if self.fresh is None:
            self.fresh = torch.zeros(self.n, dtype=torch.float32, device=self.buf.device)
        return self

    def view(self, data_ptr):
        """The fresh-gradient slice (shaped like the parameter) of the parameter stored at `data_ptr`, or None."""
        if self.fresh is None:
            return None
        m = self._fresh_views
        if m is None or data_ptr not in m:   # (parameters are re-pointed once, by FlatAdam: map them as they are now)
            m = {p.data_ptr(): self.fresh[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)}
            self._fresh_views = m
        return m.get(data_ptr)


class FlatAdam:
    """torch.optim.Adam(lr, weight_decay) (run.py:344) as ONE kernel per step (fitgnn_adam_step_f32) over a flat parameter
    buffer: the model's parameters are re-pointed to views of it, laid out like FlatGrads' gradient buffer.  torch's
    multi-tensor Adam takes 44 us for this model's six tensors, this 6.  The step counter lives on the device.
    state_dict() / load_state_dict() speak torch.optim.Adam's format (run.py keeps one optimiser across the Gc and Gs
    phases of Gc_train_2_Gs_train)."""

    def __init__(self, flat, lr=0.01, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.wd, self.betas, self.eps = flat, float(lr), float(weight_decay), betas, float(eps)
        dev = flat.buf.device
        self.P = torch.zeros_like(flat.grads)
        for p, off in zip(flat.params, flat.offsets):
            view = self.P[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.m, self.v = torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.step_count = torch.zeros(2, dtype=torch.float32, device=dev)   # [step count, fitgnn_adam_step_acc_f32's ticket word]
        self.seed_bank = None   # an ops.SeedBank this optimiser's kernel advances after every step (captured step sequences)

    def step(self):
        from . import _lib

        b = self.flat.grads
        if self.flat.fresh is not None or self.seed_bank is not None:
            # one launch: gradients of this backward (the fresh buffer) folded into the accumulated ones and cleared, the update, the
            # step count, the next step's dropout seeds
            sb = self.seed_bank
            fresh = self.flat.fresh
            _lib.check(_lib.lib().fitgnn_adam_step_acc_f32(_lib.dptr(self.P), _lib.dptr(b), _lib.dptr(fresh), _lib.dptr(self.m),
                                                           _lib.dptr(self.v), int(b.numel()), self.lr, self.betas[0], self.betas[1],
                                                           self.eps, self.wd, _lib.dptr(self.step_count),
                                                           _lib.dptr(sb.seeds) if sb is not None else None,
                                                           int(sb.seeds.numel()) if sb is not None else 0,
                                                           (sb.GOLD & 0xFFFFFFFFFFFFFFFF) if sb is not None else 0,
                                                           _lib.stream_ptr(b.device)), "fitgnn_adam_step_acc_f32")
            if sb is not None:
                sb.cursor = 0
            return
        _lib.check(_lib.lib().fitgnn_adam_step_f32(_lib.dptr(self.P), _lib.dptr(b), _lib.dptr(self.m), _lib.dptr(self.v), int(b.numel()),
                                                   self.lr, self.betas[0], self.betas[1], self.eps, self.wd, _lib.dptr(self.step_count),
                                                   _lib.stream_ptr(b.device)), "fitgnn_adam_step_f32")

    def zero_grad(self, set_to_none=False):
        self.flat.zero()

    def _views(self, buf):
        return [buf[off:off + p.numel()].view_as(p) for p, off in zip(self.flat.params, self.flat.offsets)]

    def state_dict(self):
        st = {}
        if float(self.step_count[0]) > 0:
            for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
                st[i] = {"step": self.step_count[0].clone(), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd, amsgrad=False, maximize=False,
                     params=list(range(len(self.flat.params))))
        return {"state": st, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.wd, self.eps, self.betas = float(g["lr"]), float(g["weight_decay"]), float(g["eps"]), tuple(g["betas"])
        self.m.zero_(); self.v.zero_(); self.step_count.zero_()
        for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
            if i in sd["state"]:
                e = sd["state"][i]
                m.copy_(e["exp_avg"]); v.copy_(e["exp_avg_sq"])
                self.step_count[0] = float(e["step"])


def broadcast_parameters(model, process_group=None, flat=None):
    """Data parallel: make rank 0's weights (and buffers) everyone's.  The replicated optimiser step keeps the replicas identical
    only if they START identical, and nothing else guarantees that: an unseeded run (--seed defaults to None, main.py:64 as in the
    reference) draws different initial weights on every rank.  flat: a FlatAdam whose flat parameter buffer the model's
    parameters are views of (one broadcast instead of one per tensor).  No-op without an initialised group of > 1 ranks."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) <= 1:
        return
    src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
    with torch.no_grad():
        if flat is not None:
            dist.broadcast(flat.P, src, group=process_group)
        else:
            for p in model.parameters():
                if p.data.is_contiguous():
                    dist.broadcast(p.data, src, group=process_group)
                else:   # a collective ships memory as it lies: go through a row-major copy
                    c = p.data.contiguous()
                    dist.broadcast(c, src, group=process_group)
                    p.data.copy_(c)
        for b in model.buffers():
            dist.broadcast(b.data, src, group=process_group)


def _drop_stale_sink(model):
    """A model that an earlier trainer left writing its weight gradients to THAT trainer's fresh buffer (OpConfig.grad_sink) must
    not keep doing so under a new gradient buffer."""
    cfg = getattr(model, "op_config", None)
    if cfg is not None and getattr(cfg, "grad_sink", None) is not None and hasattr(model, "set_op_config"):
        model.set_op_config(cfg.replace(grad_sink=None))


def _make_adam(model, flat, lr, weight_decay):
    """Adam(lr, weight_decay) (run.py:344) for a trainer: on the GPU ONE kernel over the flat parameter buffer (FlatAdam: same
    arithmetic as torch.optim.Adam, step counter on the device: safe to capture in a hipGraph; torch's multi-tensor Adam takes 39 us
    per step on these six tensors, this 6); on the CPU (host-side logic tests) torch's optimiser itself."""
    if next(model.parameters()).is_cuda:
        return FlatAdam(flat, lr=lr, weight_decay=weight_decay)
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)


class GDTrainer:
    def __init__(self, model, batch, lr=0.01, weight_decay=5e-4, reduction="mean", process_group=None, dedup=True,
                 task="node_cls", prune_unused_rows=False, op_config=None, global_train_count=None, lean_step=True, capture="auto"):
        """task 'node_cls': NLLLoss on log-probabilities (run.py:341); 'node_reg': L1Loss on the [n, 1] outputs (run.py:518).
        op_config (ops.OpConfig): the switches this trainer's kernels run under (set on the model; default: the model's own).
        prune_unused_rows: evaluate the last layer only on the rows that reach the loss (train nodes are own nodes of their
        cluster; the reference computes and then discards every other output, run.py:193-204).  Same loss and gradients; with a GCN
        last layer the step is the default one with the forward aggregation A_hat[loss rows, :] h instead of A_hat h over every
        row (ops.FusedGCNLastLayerRows, fwd_sub); other layers: the row-subset path over the own rows.  Off by default: bench.py's
        metric counts every non-zero of A_hat in all four SpMMs.
        global_train_count: the number of train rows of the WHOLE job when it is known up front (bench.py --shard: one rank of an
        N-rank job stepped alone); default: this batch's count, summed over the process group.
        lean_step (single rank, GPU): the step's weight gradients go straight to a buffer the optimiser kernel folds in
        (ops.OpConfig.grad_sink, fitgnn_adam_step_acc_f32: no `grad += new` launch per tensor, step count advanced by the same launch).
        capture ("auto" | True | False; single rank, GPU, fused loss): the whole step -- forward, loss, backward, Adam -- replayed from a
        hipGraph captured at the first step (dropout seeds device-resident, moved on by the optimiser kernel).  "auto": when the union is
        small enough that a second, private copy of the step's intermediates is cheap (rows x 2 KiB <= 1 GiB: the launch-bound regime,
        where ten host gaps and twenty launch floors are 15 % of a 1.1-ms S-pubmed step); a step run while the config carries profiling
        hooks (cfg.profile / profile_gemm / profile_fused) runs eagerly."""
        self.model, self.batch, self.task = model, batch, task
        if op_config is not None:
            model.set_op_config(op_config)
        from . import ops as _ops
        self.cfg = getattr(model, "op_config", _ops.DEFAULT)
        self.sub = None
        # first layer on the de-duplicated feature table when the batch carries one (same arithmetic, fewer FLOPs)
        self.dedup = dedup and getattr(batch, "row_index", None) is not None
        self.flat = FlatGrads(model.parameters())
        _drop_stale_sink(model)
        if next(model.parameters()).is_cuda:   # one kernel over the flat buffers (same arithmetic as torch.optim.Adam)
            self.opt = FlatAdam(self.flat, lr=lr, weight_decay=weight_decay)
        else:                                  # host-side logic tests (gloo): the reference's optimiser itself
            self.opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
        self.reduction = reduction
        self.pg = process_group
        self.dist = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                  and torch.distributed.get_world_size() > 1)
        if self.dist:   # replicas must start from ONE set of weights (whatever seeds the ranks drew theirs from)
            broadcast_parameters(model, process_group, flat=self.opt if isinstance(self.opt, FlatAdam) else None)
        from . import network as _net
        self.fused_loss = (task == "node_cls" and isinstance(model, _net.Classify_node) and next(model.parameters()).is_cuda)
        self.lean = bool(lean_step) and not self.dist and isinstance(self.opt, FlatAdam) and hasattr(model, "set_op_config")
        if self.lean:
            self.cfg = self.cfg.replace(grad_sink=self.flat.enable_fresh())
            model.set_op_config(self.cfg)
        small = int(getattr(batch, "n_rows", 0) or 0) * 2048 <= (1 << 30)
        from . import nn as _fnn0
        gcn_only = all(isinstance(c, _fnn0.GCNConv) for c in getattr(model, "conv", [])) and len(getattr(model, "conv", [])) > 0
        # (a model with a `logits` method -- network.APPNPNet -- gets the same fused loss on its train rows)
        self.fused_logits = (task == "node_cls" and not self.fused_loss and hasattr(model, "logits") and next(model.parameters()).is_cuda)
        # the whole step replayed from a hipGraph when it is launch-bound (a small union): the GCN models (dropout seeds on the device,
        # ops.SeedBank) and the `logits` models (APPNPNet: torch's own dropout, whose generator torch's graphs advance per replay)
        self.capture = (self.lean and ((self.fused_loss and gcn_only) or self.fused_logits)
                        and (capture is True or (capture == "auto" and small)))
        self._graph, self._graph_loss, self._bank = None, None, None
        self._y_train = batch.y.index_select(0, batch.train_idx) if (self.fused_loss or self.fused_logits) else None
        self._train_arange = None
        self.prune_forward = False
        if prune_unused_rows and self.fused_loss and batch.graph is not None:
            from . import nn as _fnn
            from .csr import RowSubset
            last = model.conv[-1] if len(model.conv) else None
            if isinstance(last, _fnn.GCNConv) and self.cfg.last_layer_on_loss_rows:
                # the aggregate-first last layer with its forward aggregation on the loss rows alone (ops.FusedGCNLastLayerRows,
                # fwd_sub): everything else of the step -- compact backward, two-hop pass -- is the full step's
                self.prune_forward = True
                self.sub = RowSubset(batch.graph, batch.train_idx)   # (also the trainer's record of what is aggregated: bench.py counts its entries)
                batch.graph._rows_fwd = (batch.train_idx, batch.train_idx._version, self.sub)   # what embed_and_head looks up
            elif isinstance(last, _fnn.GCNConv):
                core_rows = torch.nonzero(batch.core).flatten()
                self.sub = RowSubset(batch.graph, core_rows)
                pos = torch.full((batch.n_rows,), -1, dtype=torch.int64, device=core_rows.device)
                pos[core_rows] = torch.arange(core_rows.numel(), device=core_rows.device)
                self._train_pos = pos[batch.train_idx]          # train rows are own nodes (utils.py:695-698)
                assert self._train_pos.numel() == 0 or int(self._train_pos.min()) >= 0
        count = torch.tensor([float(batch.train_idx.numel())], device=self.flat.buf.device)
        if self.dist:
            torch.distributed.all_reduce(count, group=self.pg)
        self.global_count = float(count.item()) if global_train_count is None else float(global_train_count)
        # Data parallel: the gradients of everything above the first layer are complete when the first layer's backward
        # starts -- all-reduce that part of the flat buffer asynchronously (RCCL's own stream) under the first layer's
        # backward kernels, the first layer's part afterwards.  xGMI is point-to-point: two latency-bound ~1 MB calls, no
        # finer bucketing.
        self._work, self._split = None, 0
        self.comm_events = None
        if self.dist:
            first = [p for n, p in model.named_parameters() if n.startswith("conv.0.")]
            ids = {id(p) for p in first}
            if first and all(id(p) in ids for p in self.flat.params[:len(first)]) and len(first) < len(self.flat.params):
                self._split = self.flat.offsets[len(first)]
                self._late = [p for p in self.flat.params[len(first):]]
                self._pending = 0
                for p in self._late:
                    p.register_post_accumulate_grad_hook(self._grad_ready)

    def _grad_ready(self, _p):
        self._pending -= 1
        if self._pending == 0:   # the late bucket ends with the loss slot of the flat buffer
            self._work = torch.distributed.all_reduce(self.flat.buf[self._split:], group=self.pg, async_op=True)

    def _reduce_grads(self):
        """Finish the gradient all-reduce: the early bucket was launched from the hook, the first layer's follows here.
        comm_events (a list, or None): HIP-event pairs around this call on the compute stream -- the part of the all-reduce the
        step actually waits for (the early bucket runs on the collective's own stream under the first layer's backward)."""
        ev = None
        if self.comm_events is not None and self.flat.buf.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._reduce_grads_now()
        if ev is not None:
            ev[1].record()
            self.comm_events.append(ev)

    def _reduce_grads_now(self):
        if self._split and self._work is not None:
            torch.distributed.all_reduce(self.flat.buf[:self._split], group=self.pg)
            self._work.wait()
            self._work = None
        else:
            torch.distributed.all_reduce(self.flat.buf, group=self.pg)  # one RCCL all-reduce per step

    def _backward_and_step(self, loss):
        """loss = this rank's share (its sum-loss x 1 / global count): backward, gradient all-reduce with the loss riding in
        the buffer's tail slot, replicated Adam.  Returns the GLOBAL loss (the sum of the ranks' shares)."""
        self.local_loss = loss.detach()   # this rank's share of the step's loss (the whole loss without data parallelism)
        if self.dist:
            self.flat.tail[:1].copy_(loss.detach().view(1))   # before backward: the late bucket leaves from a backward hook
        loss.backward()
        if self.dist:
            self._reduce_grads()
        self.opt.step()
        return self.flat.tail[0].clone() if self.dist else loss.detach()

    def _capture_step(self):
        """Capture one step in a hipGraph (weights / optimiser state are put back after the warm-up steps)."""
        from . import ops

        dev = self.flat.buf.device
        bank = ops.SeedBank(max(len(getattr(self.model, "conv", [])), 1), dev)
        saved_m = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        saved_o = (self.opt.m.clone(), self.opt.v.clone(), self.opt.step_count.clone())
        self.model.set_op_config(self.cfg.replace(seed_bank=bank))
        self.opt.seed_bank = bank
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with _accumulate_stream_guard() as guard, torch.cuda.stream(side):
                for _ in range(2):
                    bank.cursor = 0
                    self._step_eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.model.load_state_dict(saved_m)
            self.opt.m.copy_(saved_o[0]); self.opt.v.copy_(saved_o[1]); self.opt.step_count.copy_(saved_o[2])
            if guard.mismatch:   # an older autograd graph keeps the parameters' AccumulateGrad nodes on another stream: no capture
                self.capture = False
                self.opt.seed_bank = None
                return
            g = torch.cuda.CUDAGraph()
            self._graph_loss = torch.zeros((), device=dev)
            with torch.cuda.graph(g):
                bank.cursor = 0
                self._graph_loss.copy_(self._step_eager())
            self._graph, self._bank = g, bank
        finally:
            self.model.set_op_config(self.cfg)

    def step(self):
        """One GD epoch (run.py:177-215).  Returns the global loss (over every rank's subgraphs) as a 0-dim device tensor."""
        if (self.capture and (self.fused_loss or self.fused_logits) and self.cfg.profile is None and self.cfg.profile_gemm is None
                and self.cfg.profile_fused is None):
            if self._graph is None:
                self._capture_step()
            if self._graph is not None:
                self._graph.replay()
                self.local_loss = self._graph_loss
                return self._graph_loss   # (the step's static loss buffer: the next replay overwrites it)
        return self._step_eager()

    def _step_eager(self):
        m, b = self.model, self.batch
        m.train()
        self.flat.zero()  # optimizer.zero_grad(); grads live in the flat buffer
        if self._split:
            self._pending, self._work = len(self._late), None
        scale = 1.0 / self.global_count if self.reduction == "mean" else 1.0
        if self.fused_loss:   # logits -> loss and d(loss)/d(logits) in one kernel (same arithmetic as log_softmax + NLLLoss)
            from .ops import SoftmaxNLL
            if self.sub is not None and not self.prune_forward:
                z = (m.embed_and_head(b.x_table, b.edge_index, b.row_index, out_rows=self.sub) if self.dedup
                     else m.embed_and_head(b.x, b.edge_index, out_rows=self.sub))
                loss = SoftmaxNLL.apply(z, self._train_pos, self._y_train, scale)
            else:
                z = (m.embed_and_head(b.x_table, b.edge_index, b.row_index, loss_rows=b.train_idx, compact_logits=True,
                                      forward_rows_only=self.prune_forward) if self.dedup
                     else m.embed_and_head(b.x, b.edge_index, loss_rows=b.train_idx, compact_logits=True, forward_rows_only=self.prune_forward))
                if z.shape[0] == b.train_idx.numel() and z.shape[0] != b.n_rows:   # the logits of the train rows only, in their order
                    if self._train_arange is None:
                        self._train_arange = torch.arange(z.shape[0], dtype=torch.int64, device=z.device)
                    rows = self._train_arange
                else:
                    rows = b.train_idx
                if self.lean and not self.dist:   # the loss's own gradient handed to backward (no ones-fill, no multiplication by it)
                    from .ops import softmax_nll_raw
                    loss1, dz = softmax_nll_raw(z, rows, self._y_train, scale)
                    self.local_loss = loss1[0]
                    z.backward(dz)
                    self.opt.step()
                    return loss1[0]
                loss = SoftmaxNLL.apply(z, rows, self._y_train, scale)
            return self._backward_and_step(loss)
        if self.fused_logits:
            from .ops import SoftmaxNLL
            z = m.logits(b.x_table, b.edge_index, x_index=b.row_index) if self.dedup else m.logits(b.x, b.edge_index)
            if self.lean and not self.dist:   # the loss's own gradient handed to backward (no ones-fill, no multiplication by it)
                from .ops import softmax_nll_raw
                loss, dz = softmax_nll_raw(z, b.train_idx, self._y_train, scale)
                self.local_loss = loss[0]
                z.backward(dz)
                self.opt.step()
                return loss[0]
            return self._backward_and_step(SoftmaxNLL.apply(z, b.train_idx, self._y_train, scale))
        out = m(b.x_table, b.edge_index, x_index=b.row_index) if self.dedup else m(b.x, b.edge_index)
        sel = out.index_select(0, b.train_idx)
        if self.task == "node_reg":
            loss_sum = F.l1_loss(sel.view(-1, 1), b.y.index_select(0, b.train_idx).view(-1, 1), reduction="sum")
        else:
            loss_sum = F.nll_loss(sel, b.y.index_select(0, b.train_idx), reduction="sum")
        return self._backward_and_step(loss_sum * scale)

    @torch.no_grad()
    def evaluate(self, mask_idx):
        self.model.eval()
        b = self.batch
        out = self.model(b.x, b.edge_index)
        sel = out.index_select(0, mask_idx)
        y = b.y.index_select(0, mask_idx)
        return F.nll_loss(sel, y), (sel.argmax(1) == y).float().mean()


class _CapturedSteps:
    """Launch-bound step sequences (one optimiser step per small batch) replayed from hipGraphs: each batch's
    forward + loss + backward + Adam is captured once; dropout seeds live on the device (ops.SeedBank) and are advanced by
    a kernel inside every captured step.  Users provide self.model / self.opt (capturable Adam) / self.flat and
    _steps() (the batches) and _one(batch) (one eager step returning the detached loss)."""

    def _build_graphs(self):
        """Capture the steps in hipGraphs of `steps_per_graph` (8) consecutive batches each (shared memory pool: the graphs replay one
        after another, in order)."""
        from . import ops

        dev = self.flat.buf.device
        self._bank = ops.SeedBank(max(len(self.model.conv), 1), dev)
        self._losses = torch.zeros(len(self._steps()), device=dev)
        # warm-up on a side stream (library workspaces, autograd buffers), then put the weights / Adam state back
        saved_m = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        # Adam's moment / step tensors must EXIST before capture (a lazily initialised state would be allocated and
        # zeroed inside the first captured step, i.e. reset on every replay): warm up, then restore them in place
        flat_opt = isinstance(self.opt, FlatAdam)   # its state (m, v, the device-resident step counter) exists from construction
        if flat_opt:
            saved_o = (self.opt.m.clone(), self.opt.v.clone(), self.opt.step_count.clone())
        else:
            saved_o = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.opt.state.items()}
        # the captured kernels read their dropout seeds through the bank's device pointers: the model runs under a copy of
        # its config that carries the bank while the steps are built (replays re-run no Python)
        prev = self.model.op_config
        self.model.set_op_config(prev.replace(seed_bank=self._bank))
        # the optimiser kernel moves the seeds on after every step (one launch less per step) when it is the flat one
        adam_advances = flat_opt and getattr(self, "lean", True)
        if adam_advances:
            self.opt.seed_bank = self._bank
        # what a step keeps per batch (A_hat x of a narrow first layer, ops.aggregated_input) is formed NOW: made lazily inside a
        # step it would be captured with it and replayed every epoch
        prepare = getattr(self.model, "prepare_static", None)
        if prepare is not None:
            for b in self._steps():
                if b is not None:
                    prepare(*self._static_inputs(b))
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with _accumulate_stream_guard() as guard, torch.cuda.stream(side):
                for b in self._steps()[:2]:
                    if not adam_advances:
                        self._bank.advance()
                    self._one(b)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.model.load_state_dict(saved_m)
            if flat_opt:
                self.opt.m.copy_(saved_o[0]); self.opt.v.copy_(saved_o[1]); self.opt.step_count.copy_(saved_o[2])
            else:
                for p, st in self.opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(saved_o[p][k]) if p in saved_o and k in saved_o[p] else v.zero_()
            self.flat.zero()
            if guard.mismatch:   # (see _accumulate_stream_guard) the steps run eagerly
                self.capture = False
                if adam_advances:
                    self.opt.seed_bank = None
                return
            pool = torch.cuda.graph_pool_handle()
            self._graphs = []
            steps = self._steps()
            per = max(int(getattr(self, "steps_per_graph", 8)), 1)   # consecutive steps share a hipGraph: a replay costs ~9 us of its own
            for k0 in range(0, len(steps), per):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    for k in range(k0, min(k0 + per, len(steps))):
                        if not adam_advances:
                            self._bank.advance()
                        self._bank.cursor = 0
                        slot = self._losses[k:k + 1]
                        r = self._one(steps[k], loss_out=slot)   # (a loss kernel that can write the slot itself saves the copy)
                        if r.data_ptr() != slot.data_ptr():
                            slot.copy_(r.view(1))
                self._graphs.append(g)
            # capturing does not execute: the weights are untouched, but the gradient buffer was only zeroed eagerly
            self.flat.zero()
        finally:
            self.model.set_op_config(prev)


    def _replay(self):
        """The epoch's losses from the captured steps, or None when the steps cannot be captured (the caller then runs them eagerly)."""
        if self._graphs is None:
            self._build_graphs()
            if self._graphs is None:
                return None
        self.flat.zero()
        for g in self._graphs:
            g.replay()
        return self._losses


class MBTrainer(_CapturedSteps):
    """node_train_Gs_MB (run.py:217-252): per epoch ONE zero_grad, then for every loader batch (128 subgraphs,
    run.py:336) that holds a train node: forward, loss over the batch's train nodes, backward, optimiser step.
    The reference never clears the gradients between batches, so batch k steps with the SUM of the gradients of
    batches 0..k of this epoch; reproduced here because it changes the trained weights.
    Sequential by construction: single GPU only."""

    def __init__(self, model, batch, batch_size=128, lr=0.01, weight_decay=5e-4, reduction="mean", capture=False):
        from .csr import CSRGraph, register

        self.model, self.reduction = model, reduction
        self.capture, self._graphs = bool(capture), None
        self.flat = FlatGrads(model.parameters())
        _drop_stale_sink(model)
        self.opt = _make_adam(model, self.flat, lr, weight_decay)
        ei = batch.edge_index
        order = torch.argsort(ei[0], stable=True)
        src_sorted = ei[0][order].contiguous()
        self.parts = []
        self.n_loader_batches = 0
        for r0, r1 in batch.slice_batches(batch_size):
            self.n_loader_batches += 1
            tm = batch.train_mask[r0:r1]
            if not bool(tm.any()):
                continue  # run.py:225 `if True in train_mask`
            lo, hi = (int(v) for v in torch.searchsorted(src_sorted, torch.tensor([r0, r1], device=ei.device)))
            e = (ei[:, order[lo:hi]] - r0).contiguous()
            if e.is_cuda:
                register(e, CSRGraph(e, r1 - r0, mode="gcn", ptr=batch.ptr[(batch.ptr >= r0) & (batch.ptr <= r1)] - r0), "gcn")
            self.parts.append((batch.x[r0:r1], e, batch.y[r0:r1].index_select(0, torch.nonzero(tm).flatten()),
                               torch.nonzero(tm).flatten()))
        self.n_train = sum(int(p[3].numel()) for p in self.parts)

    def _steps(self):
        return self.parts

    def _static_inputs(self, part):
        return part[0], part[1]

    def _one(self, part, loss_out=None):
        x, e, y, idx = part
        loss = F.nll_loss(self.model(x, e).index_select(0, idx), y, reduction=self.reduction)
        loss.backward()           # accumulates into the flat buffer: no zero_grad between batches (run.py:222)
        self.opt.step()
        return loss.detach()

    def step(self):
        self.model.train()
        denom = self.n_loader_batches if self.reduction == "mean" else max(self.n_train, 1)
        if self.capture and self.flat.buf.is_cuda:
            losses = self._replay()
            if losses is not None:
                return losses.sum() / denom
        self.flat.zero()
        total = torch.zeros((), device=self.flat.buf.device)
        for part in self.parts:
            total += self._one(part)
        return total / denom


class GraphTrainer(_CapturedSteps):
    """graph_train_Gs / graph_train_Gc (run.py:254-269, :288-304) on a fitgnn_amd.graph_data.GraphSet.

    Per epoch ONE zero_grad, then per batch of `batch_size` graphs: forward, loss, backward, optimiser step -- the
    gradients are never cleared between batches (quirk i), and regression targets go through `.type(torch.long)`
    (run.py:260,:294: values are truncated toward zero; quirk ii) -- both reproduced.  Batches are contiguous ranges
    of `order` (a fixed permutation of the split's graphs; the reference reshuffles its loader every epoch, here the
    order is drawn once so that every batch's CSR is built once and stays resident).
    kind 'gs': model(set_gs, batch_tensor) pools the masked rows of the subgraph union; kind 'gc': model(gc)."""

    def __init__(self, model, gset, graphs, kind="gs", batch_size=128, lr=0.01, weight_decay=5e-4, task="graph_reg",
                 multi_prop=True, prop=0, truncate_targets=True, capture=False, share=None, rank=None, world=None,
                 process_group=None, batches=None, global_sizes=None, accumulate=True, reshuffle=False, lean_step=True):
        """capture=True: every batch step (forward, loss, backward, Adam) is captured once in a hipGraph and replayed
        -- the steps are launch-bound (small batches, ~40 kernels each).  Dropout seeds then live on the device
        (ops.SeedBank) and are advanced by a kernel inside each captured step.
        Data parallel (graph level): under an initialised process group every rank takes graphs
        ids[rank::world] of EVERY global batch (the unit is a whole graph: pooling is intra-graph, nothing is
        exchanged in the forward pass), losses are sums scaled by 1 / global batch size, and the accumulated gradient
        buffer is all-reduced once per step (see _dp_step for how the never-cleared gradients stay exact).
        batches (+ global_sizes): pre-built batch dicts (tests / custom pipelines) instead of gset + graphs.
        accumulate=False: clear the gradients before every batch (the baselines' loops, run.py:988-991, :1058-1060).
        reshuffle=True: re-draw the graph order before every epoch, as the reference's DataLoader(shuffle=True) does
        (run.py:710).  With capture=True the batches are assembled on the device into fixed-capacity buffers and ONE captured step
        is replayed for every full batch (graph_data.PaddedBatchPlan; S-qm9: 0.13 s per epoch); without it -- or where the plan
        does not apply: a first layer that is not a GCNConv on a narrow input, data parallelism -- every batch's CSR is rebuilt on
        the host and the steps run eagerly (≈ 3 ms per batch).  reshuffle="auto": reshuffle only where the plan applies, else
        the order is drawn once and each batch's captured step replayed (the default, reshuffle=False).
        lean_step=False (A/B): the step as round 3 had it -- autograd's own `grad += new` per tensor, loss.backward() from a ones
        fill, the seeds and the step count advanced by launches of their own."""
        import types

        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = int(world if world is not None else (torch.distributed.get_world_size(process_group) if dist_on else 1))
        self.rank = int(rank if rank is not None else (torch.distributed.get_rank(process_group) if dist_on else 0))
        self.pg = process_group

        self.model, self.kind, self.task, self.multi_prop, self.prop = model, kind, task, multi_prop, prop
        self.truncate = truncate_targets
        self.lean = bool(lean_step)
        self.accumulate = bool(accumulate)
        self.capture, self._graphs = bool(capture), None
        if share is not None:   # evaluation-only views of the same model: one optimiser / gradient buffer (run.py:718-719)
            self.opt, self.flat = share.opt, share.flat
        else:
            self.flat = FlatGrads(model.parameters())
            self.opt = _make_adam(model, self.flat, lr, weight_decay)
        self._rebuild, self._plan, self._shuffled_graph = None, None, None
        if batches is None and reshuffle:
            want_capture = self.capture
            if want_capture:
                self._plan = self._make_plan(gset, kind, batch_size, lean_step, share)
            if reshuffle != "auto" or self._plan is not None:   # "auto": reshuffle only where it replays a captured step
                self.capture = False
                self._rebuild = (gset, [int(g) for g in graphs], kind, batch_size, types)
        if batches is not None:   # pre-built (entries may be None: this rank holds no graph of that batch)
            self.batches = list(batches)
            self.global_sizes = list(global_sizes) if global_sizes is not None else [int(b["y"].shape[0]) for b in self.batches]
        else:
            graphs = [int(g) for g in graphs]
            self.batches, self.global_sizes = [], []
            # contiguous runs of graph ids inside `graphs` are merged into ranges; a batch = list of ranges
            for b0 in range(0, len(graphs), batch_size):
                ids_all = graphs[b0:b0 + batch_size]
                ids = ids_all[self.rank::self.world]
                self.global_sizes.append(len(ids_all))
                if not ids:
                    self.batches.append(None)   # this rank holds no graph of a short last batch: it still joins the step
                    continue
                pieces = [gset.batch_ids(ids, kind)] if not _is_range(ids) else [gset.batch(ids[0], ids[-1] + 1, kind)]
                self.batches.append(_cat_pieces(pieces, kind, types))
        if self.world > 1:
            self.capture = False   # the per-step collective is issued eagerly
        if share is None:
            if self.lean and self.world == 1 and isinstance(self.opt, FlatAdam) and hasattr(model, "set_op_config"):
                # weight gradients straight into a buffer the optimiser kernel folds in (no `grad += new` launch per tensor); not
                # under data parallelism, whose all-reduce runs over the accumulated buffer between backward and step
                self.flat.enable_fresh()
                model.set_op_config(model.op_config.replace(grad_sink=self.flat))
            else:
                _drop_stale_sink(model)
        if self.task == "graph_reg":   # static per batch: formed now, not inside a (captured) step
            for b in self.batches:
                if b is not None and "_tgt" not in b:
                    b["_tgt"] = self._target(b["y"])

    def _target(self, y):
        """The regression target of a batch as the reference forms it: y.type(torch.long) (run.py:260,:294: truncation), property
        column `prop` of a multi-property dataset, as a float column."""
        if self.truncate:
            y = y.long()
        return (y[:, self.prop].view(-1, 1) if self.multi_prop else y).float().contiguous()

    def _loss(self, out, y, reduction="mean", b=None):
        if self.task == "graph_reg":
            # the target is static per batch: formed once (a captured step would otherwise replay its cast / gather / copy kernels)
            if b is None:
                tgt = self._target(y)
            else:
                if "_tgt" not in b:
                    b["_tgt"] = self._target(y)
                tgt = b["_tgt"]
            if out.is_cuda and out.dtype == torch.float32 and tgt.shape == out.shape:
                from .ops import L1Loss
                return L1Loss.apply(out, tgt, (1.0 / max(out.numel(), 1)) if reduction == "mean" else 1.0)
            return F.l1_loss(out, tgt.to(out.dtype), reduction=reduction)
        if self.truncate:
            y = y.long()
        # graph_cls: CrossEntropyLoss on the model's SOFTMAX output (run.py:583 on network.py:94,133: a double softmax, kept)
        return F.cross_entropy(out, y.long().flatten(), reduction=reduction)

    def _dp_step(self, k, b):
        """One data-parallel batch step.  The reference never clears the gradients inside an epoch, so after step k the
        buffer must hold G(k) = sum over steps <= k and over ranks of the local gradients.  The buffer enters step k
        holding G(k-1) on every rank: scale it by 1 / world, add this rank's gradient of (local loss sum / global
        batch size), all-reduce(sum): sum_r (G(k-1)/world + g_r(k)) = G(k-1) + sum_r g_r(k) = G(k)."""
        buf = self.flat.buf
        if k > 0:
            buf.mul_(1.0 / self.world)
        loss = torch.zeros((), device=buf.device)
        if b is not None:
            loss = self._loss(self._forward(b), b["y"], reduction="sum", b=b) / float(self.global_sizes[k])
            loss.backward()
        torch.distributed.all_reduce(buf, group=self.pg)
        self.opt.step()
        return loss.detach()

    def _forward(self, b):
        if self.kind == "gs":
            return self.model(b, b["graph_of_masked"])
        return self.model(b["gc"])   # 'gc' and 'orig': one block-diagonal graph batch

    def _steps(self):
        return self.batches

    def _static_inputs(self, b):
        if self.kind == "gs":
            return b["x"], b["edge_index"], b.get("mask_idx")
        return b["gc"].x, b["gc"].edge_index

    def _one(self, b, loss_out=None):
        """One batch step.  loss_out (a one-element device tensor, optional): where the loss may be written directly."""
        if not self.accumulate:
            self.flat.zero()
        out = self._forward(b)
        tgt = b.get("_tgt") if self.task == "graph_reg" else None
        if self.lean and tgt is not None and out.is_cuda and out.dtype == torch.float32 and tgt.shape == out.shape and out.requires_grad:
            # L1 loss and its gradient from one launch, handed to autograd as the output's gradient: loss.backward() would fill a
            # one, multiply the stored gradient by it and copy the loss to its slot
            from .ops import l1_loss_raw
            loss, grad = l1_loss_raw(out, tgt, 1.0 / max(out.numel(), 1), loss_out)
            out.backward(grad)
            self.opt.step()
            return loss.view(())
        loss = self._loss(out, b["y"], b=b)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _make_plan(self, gset, kind, batch_size, lean_step, share):
        """reshuffle=True with capture=True: batches assembled on the device into fixed-capacity buffers (graph_data.PaddedBatchPlan)
        and ONE captured step replayed for every full batch of every epoch.  Needs what makes the step independent of the batch's
        contents: the subgraph view, a first GCN layer on the aggregated narrow input (its A_hat x is gathered like everything else),
        the L1 regression step whose loss kernel writes a device slot, a single rank.  Otherwise None: the eager rebuild."""
        from . import nn as fnn
        from . import ops
        from .graph_data import PaddedBatchPlan

        model = self.model
        ok = (kind in ("gs", "gc") and lean_step and share is None and self.world == 1 and self.task in ("graph_reg", "graph_cls")
              and isinstance(self.opt, FlatAdam) and gset is not None and gset.x.is_cuda and getattr(model, "num_layers", 0) > 0
              and isinstance(model.conv[0], fnn.GCNConv) and 1 <= int(batch_size) <= 1024)
        if not ok:
            return None
        x_all = gset.gs_x if kind == "gs" else gset.gc_x
        probe = torch.zeros((1, x_all.shape[1]), dtype=torch.float32, device=gset.x.device)
        if not ops.narrow_input_supported(probe, model.conv[0].lin.weight, model.op_config):
            return None
        target_fn = self._target if self.task == "graph_reg" else (lambda y: y.float())
        try:
            return PaddedBatchPlan(gset, batch_size, target_fn, kind=kind)
        except ValueError:   # (a union whose normalised adjacency is not symmetric)
            return None

    def _build_shuffled_graph(self):
        """Capture [assemble the batch at the device counter, forward, L1 loss, backward, Adam] once (PaddedBatchPlan.batch is static)."""
        from . import ops

        plan, dev = self._plan, self.flat.buf.device
        bank = ops.SeedBank(max(len(getattr(self.model, "conv", [])), 1), dev)
        saved_m = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        saved_o = (self.opt.m.clone(), self.opt.v.clone(), self.opt.step_count.clone())
        prev = self.model.op_config
        self.model.set_op_config(prev.replace(seed_bank=bank))
        self.opt.seed_bank = bank
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with _accumulate_stream_guard() as guard, torch.cuda.stream(side):
                for _ in range(2):   # warm-up (library workspaces, autograd buffers) on the epoch's first batch
                    plan.step_idx.zero_()
                    plan.assemble()
                    bank.cursor = 0
                    self._one(self._plan_batch(), loss_out=plan.loss_slot)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.model.load_state_dict(saved_m)
            self.opt.m.copy_(saved_o[0]); self.opt.v.copy_(saved_o[1]); self.opt.step_count.copy_(saved_o[2])
            self.flat.zero()
            plan.step_idx.zero_(); plan.loss_sum.zero_(); plan.loss_slot.zero_()
            if guard.mismatch:   # (see _accumulate_stream_guard) every batch of every epoch takes the eager way
                self._shuffled_graph = "eager"
                self.opt.seed_bank = None
                return
            pool = torch.cuda.graph_pool_handle()
            graphs = []
            for reps in (1, max(int(getattr(self, "steps_per_graph", 8)), 1)):   # one step, and a run of steps (a replay costs ~10 us of its own)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    for _ in range(reps):
                        plan.assemble()
                        bank.cursor = 0
                        r = self._one(self._plan_batch(), loss_out=plan.loss_slot)
                        if r.data_ptr() != plan.loss_slot.data_ptr():   # (a loss that is not written to the slot by its own kernel)
                            plan.loss_slot.copy_(r.detach().view(1))
                graphs.append((reps, g))
            self._shuffled_graph, self._bank = graphs, bank
            plan.step_idx.zero_(); plan.loss_sum.zero_(); plan.loss_slot.zero_()   # (capturing does not execute)
        finally:
            self.model.set_op_config(prev)

    def _plan_batch(self):
        b = self._plan.batch
        if self.task == "graph_reg" and "_tgt" not in b:
            b["_tgt"] = b["y"]   # (the plan's targets are already in the form the regression loss takes)
        return b

    def _step_shuffled(self):
        """One epoch over freshly shuffled batches: full batches that fit the plan's capacities replay the captured step, the others
        (and a last, shorter batch) run the eager way."""
        import numpy as np

        plan = self._plan
        gset, graphs, kind, batch_size, types = self._rebuild
        perm = torch.randperm(len(graphs)).tolist()   # torch's generator, as the DataLoader's sampler
        ids = np.asarray([graphs[i] for i in perm], dtype=np.int64)
        B = plan.B
        n_full = len(ids) // B
        full = ids[: n_full * B].reshape(n_full, B)
        fits = plan.fits(full) if n_full else np.zeros(0, dtype=bool)
        plan.set_epoch(full.reshape(-1))
        if self._shuffled_graph is None and n_full:
            self._build_shuffled_graph()
        self.flat.zero()
        total = torch.zeros((), device=self.flat.buf.device)

        def eager(id_list):
            b = _cat_pieces([gset.batch_ids(id_list, kind)], kind, types)
            if self.task == "graph_reg":
                b["_tgt"] = self._target(b["y"])
            return self._one(b)

        if self._shuffled_graph == "eager":
            fits = np.zeros(n_full, dtype=bool)
        (_, g_one), (per, g_run) = self._shuffled_graph if (n_full and self._shuffled_graph != "eager") else ((1, None), (1, None))
        k = 0
        while k < n_full:
            if per > 1 and k + per <= n_full and bool(fits[k:k + per].all()):
                g_run.replay()
                k += per
            elif fits[k]:
                g_one.replay()
                k += 1
            else:
                plan.step_idx.add_(1)
                total += eager(full[k].tolist())
                k += 1
        n_batches = n_full
        if len(ids) > n_full * B:
            total += eager(ids[n_full * B:].tolist())
            n_batches += 1
        self.global_sizes = [B] * n_full + ([len(ids) - n_full * B] if n_batches > n_full else [])
        total = total + plan.loss_sum[0] + plan.loss_slot[0]   # (the last replayed step's loss is still in its slot)
        return total / max(n_batches, 1)

    def _reshuffle(self):
        gset, graphs, kind, batch_size, types = self._rebuild
        perm = torch.randperm(len(graphs)).tolist()   # torch's generator, as the DataLoader's sampler; same on every rank
        graphs = [graphs[i] for i in perm]            # when the ranks share the seed
        self.batches, self.global_sizes = [], []
        for b0 in range(0, len(graphs), batch_size):
            ids_all = graphs[b0:b0 + batch_size]
            ids = ids_all[self.rank::self.world]
            self.global_sizes.append(len(ids_all))
            self.batches.append(_cat_pieces([gset.batch_ids(ids, kind)], kind, types) if ids else None)

    def step(self):
        self.model.train()
        if self._plan is not None:
            return self._step_shuffled()
        if self._rebuild is not None:
            self._reshuffle()
        if self.capture and self.flat.buf.is_cuda:
            losses = self._replay()
            if losses is not None:
                return losses.sum() / max(len(self.batches), 1)
        self.flat.zero()
        total = torch.zeros((), device=self.flat.buf.device)
        if self.world > 1:
            for k, b in enumerate(self.batches):
                total += self._dp_step(k, b)
            torch.distributed.all_reduce(total, group=self.pg)   # sum of the ranks' shares of every batch mean
            return total / max(len(self.batches), 1)
        for b in self.batches:
            total += self._one(b)
        return total / max(len(self.batches), 1)

    @torch.no_grad()
    def evaluate(self):
        """graph_infer_Gs / graph_val_Gc: mean batch loss; regression losses of the Gs path are divided by the std of
        the (truncated) labels (run.py:323-325)."""
        self.model.eval()
        total, labels = torch.zeros((), device=self.flat.buf.device), []
        for b in self.batches:
            total += self._loss(self._forward(b), b["y"], b=b)
            y = b["y"].long() if self.truncate else b["y"]
            labels.append(y[:, self.prop] if self.multi_prop else y.flatten())
        if self.task == "graph_reg" and self.kind == "gs":
            total = total / torch.cat(labels).float().std()
        return total / max(len(self.batches), 1)

    @torch.no_grad()
    def accuracy(self):
        """graph_cls accuracy over ALL graphs of the split (the reference reports the last batch's only, run.py:284,324)."""
        self.model.eval()
        hit = n = 0
        for b in self.batches:
            pred = self._forward(b).argmax(1)
            hit += int((pred == b["y"].long().flatten()).sum())
            n += int(pred.numel())
        return hit / max(n, 1)


def _is_range(ids):
    return all(b == a + 1 for a, b in zip(ids, ids[1:]))


def _cat_pieces(pieces, kind, types):
    """One batch from per-range pieces (block-diagonal concatenation; a single range needs no copy)."""
    if len(pieces) == 1:
        p = pieces[0]
        x, e, graph, mask, y = p["x"], p["edge_index"], p["graph"], p["mask"], p["y"]
    else:
        from .csr import CSRGraph, register

        offs = np_cumsum([int(p["x"].shape[0]) for p in pieces])
        goff = np_cumsum([p["n_graphs"] for p in pieces])
        x = torch.cat([p["x"] for p in pieces])
        e = torch.cat([p["edge_index"] + o for p, o in zip(pieces, offs)], 1).contiguous()
        graph = torch.cat([p["graph"] + o for p, o in zip(pieces, goff)])
        mask = torch.cat([p["mask"] for p in pieces]) if kind == "gs" else None
        y = torch.cat([p["y"] for p in pieces])
        if x.is_cuda:
            register(e, CSRGraph(e, int(x.shape[0]), mode="gcn"), "gcn")
    n_graphs = int(y.shape[0])
    b = dict(x=x, edge_index=e, mask=mask, y=y, n_graphs=n_graphs)
    if kind == "gs":
        b["mask_idx"] = torch.nonzero(mask).flatten()      # precomputed: x[mask] would synchronise with the host
        b["graph_of_masked"] = graph[mask]
        if x.is_cuda:   # the pool's segment index, built now (it synchronises once; a captured step must not)
            from . import ops
            ops.pool_index(b["graph_of_masked"], n_graphs, b["mask_idx"], int(x.shape[0]))
            ops.pool_index(b["graph_of_masked"], n_graphs, None, int(b["mask_idx"].numel()))   # (the pool over the compact pooled rows)
    else:
        b["gc"] = types.SimpleNamespace(x=x, edge_index=e, batch=graph, num_graphs=n_graphs)
        if x.is_cuda:
            from . import ops
            ops.pool_index(graph, n_graphs, None, int(x.shape[0]))
    return b


def np_cumsum(v):
    out, s = [], 0
    for a in v:
        out.append(s)
        s += a
    return out
