"""Graph-level datasets (QM9-shaped: many small graphs) as flat device-resident arrays.

The reference walks the dataset graph by graph in Python (main.py:370): coarsening_regression (utils.py:376-534: components,
coarsen, per-cluster subgraphs) then load_graph_data (utils.py:811-852: Gc = pooled features + coarse edges), and its
graph-level models loop over every subgraph of every graph of a batch (network.py:120-130, :189-204).  Here the whole
dataset is ONE block-diagonal graph: one batched contraction (coarsening.coarsen_batch), one subgraph assembly
(data.assemble_subgraphs_torch) and one pooling launch build, for all graphs at once,
    Gs : the union of all cluster subgraphs, rows ordered by (graph, cluster, node id), with the reference's row mask
    Gc : pooled features C.X, coarse edges, cluster -> graph vector
and a training batch is a contiguous range of graphs of a fixed (seeded) order.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import coarsening
from . import data as fdata
from .csr import CSRGraph, register


def synthetic_molecules(n_graphs, seed=0, n_features=11, n_targets=19):
    """QM9-shaped stand-in: node count ~ round(N(18, 3^2)) clipped to [3, 29]; a ring plus random
    chords up to ~18.7 undirected edges on average (connected); F = 11 features, 19 float targets that are smooth
    functions of the graph (size, chord count, mean feature) so that regression is learnable."""
    rng = np.random.default_rng(seed)
    n = np.clip(np.rint(rng.normal(18, 3, size=n_graphs)), 3, 29).astype(np.int64)
    node_ptr = np.zeros(n_graphs + 1, dtype=np.int64)
    np.cumsum(n, out=node_ptr[1:])
    src, dst = [], []
    chords = np.zeros(n_graphs)
    for g in range(n_graphs):
        k, o = int(n[g]), int(node_ptr[g])
        ring = np.arange(k)
        und = {(min(a, b), max(a, b)) for a, b in zip(ring, np.roll(ring, -1)) if a != b}
        want = max(len(und), int(round(k * 18.7 / 18)))
        tries = 0
        while len(und) < want and tries < 8 * want:
            a, b = rng.integers(0, k, size=2)
            tries += 1
            if a != b:
                und.add((min(a, b), max(a, b)))
        chords[g] = len(und) - k
        u = np.array(sorted(und), dtype=np.int64) + o
        src += [u[:, 0], u[:, 1]]
        dst += [u[:, 1], u[:, 0]]
    ei = np.stack([np.concatenate(src), np.concatenate(dst)])
    ei = ei[:, np.lexsort((ei[1], ei[0]))]
    N = int(node_ptr[-1])
    x = rng.random((N, n_features), dtype=np.float32)
    gid = np.repeat(np.arange(n_graphs), n)
    mean_x = np.zeros((n_graphs, n_features)); np.add.at(mean_x, gid, x); mean_x /= n[:, None]
    base = np.stack([n / 18.0, chords / 3.0, mean_x[:, 0], mean_x[:, 1]], 1)
    mix = rng.standard_normal((4, n_targets))
    # QM9-like magnitudes (units to tens): the reference casts targets to long (run.py:260,:294), so sub-unit targets
    # would all truncate to 0
    y = (8.0 * (base @ mix) + 0.2 * rng.standard_normal((n_graphs, n_targets))).astype(np.float32)
    return dict(node_ptr=node_ptr, edge_index=ei, x=x, y=y)


def synthetic_graph_classes(n_graphs, seed=0, n_features=3, n_classes=2, mean_nodes=19):
    """TU-dataset-shaped stand-in (PROTEINS: 1113 graphs, ~19 nodes (csv shows halved counts), 3 node labels, 2
    classes; dataset_info.csv:11): the class sets the chord density and the node-label distribution, so both structure and
    features carry signal.  Same dict as synthetic_molecules with integer class labels in y [G, 1]."""
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, n_classes, size=n_graphs)
    n = np.clip(np.rint(rng.normal(mean_nodes, 4, size=n_graphs)), 4, 3 * mean_nodes).astype(np.int64)
    node_ptr = np.zeros(n_graphs + 1, dtype=np.int64)
    np.cumsum(n, out=node_ptr[1:])
    src, dst, xs = [], [], []
    for g in range(n_graphs):
        k, o, c = int(n[g]), int(node_ptr[g]), int(cls[g])
        ring = np.arange(k)
        und = {(min(a, b), max(a, b)) for a, b in zip(ring, np.roll(ring, -1)) if a != b}
        want = len(und) + int(round(k * (0.3 + 0.9 * c / max(n_classes - 1, 1))))
        tries = 0
        while len(und) < want and tries < 8 * want:
            a, b = rng.integers(0, k, size=2)
            tries += 1
            if a != b:
                und.add((min(a, b), max(a, b)))
        u = np.array(sorted(und), dtype=np.int64) + o
        src += [u[:, 0], u[:, 1]]
        dst += [u[:, 1], u[:, 0]]
        p = np.full(n_features, 1.0)
        p[c % n_features] += 2.0
        xs.append(np.eye(n_features, dtype=np.float32)[rng.choice(n_features, size=k, p=p / p.sum())])
    ei = np.stack([np.concatenate(src), np.concatenate(dst)])
    ei = ei[:, np.lexsort((ei[1], ei[0]))]
    return dict(node_ptr=node_ptr, edge_index=ei, x=np.concatenate(xs), y=cls.reshape(-1, 1).astype(np.float32))


class GraphSet:
    """All graphs of a graph-level dataset, coarsened and assembled once, resident on the device.

    Rows of a graph are contiguous everywhere: original nodes node_ptr, clusters cluster_ptr (Gc rows), Gs union
    rows gs_ptr.  `gs_mask` is the reference's row mask M.mask (utils.py:498-503): with --extra_node it marks the
    FIRST k rows of each sorted subgraph, k = number of own nodes -- not the own nodes themselves (SURVEY §8 a12
    quirk iii); `gs_core` marks the true own nodes.  network.*_graph_gs pools x[mask] per graph."""

    def __init__(self, mol, ratio=0.5, extra_node=False, device="cuda", spectral="dense", reference_mask=True,
                 cluster_node=False):
        dev = torch.device(device)
        node_ptr, ei = np.asarray(mol["node_ptr"]), np.asarray(mol["edge_index"])
        N, G = int(node_ptr[-1]), len(node_ptr) - 1
        self.n_graphs, self.node_ptr = G, node_ptr
        W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
        # main.py:370-377 passes Loukas r = 1 - coarsening_ratio; every graph is one connected component here
        self.co = coarsening.coarsen_batch(W, node_ptr, r=1 - ratio, device=dev, spectral=spectral)
        co = self.co
        self.cluster_ptr = np.asarray(co.cluster_off)
        n = co.n_clusters
        self.x = torch.as_tensor(mol["x"]).to(dev).float()
        self.y = torch.as_tensor(mol["y"]).to(dev).float()
        self.edge_index = torch.from_numpy(ei).to(dev)                # the original graphs (baselines, run.py:967-1100)
        self.node_graph = torch.from_numpy(np.repeat(np.arange(G), np.diff(node_ptr))).to(dev)
        # ---- Gc (load_graph_data, utils.py:811-852) ----
        self.gc_x = co.pool(self.x)
        coo = co.Wc.tocoo()
        self.gc_edge_index = torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64)).to(dev)
        self.gc_graph = torch.from_numpy(np.repeat(np.arange(G), np.diff(self.cluster_ptr))).to(dev)
        # ---- Gs (utils.py:417-534), all clusters of all graphs at once ----
        if cluster_node:   # utils.py:424-470: one new node per neighbouring cluster, carrying its pooled features
            sub = fdata.assemble_subgraphs_cluster(ei, N, co.assign, n, co.Wc)
            sub = {k: (torch.from_numpy(v).to(dev) if k != "ptr" else torch.from_numpy(v)) for k, v in sub.items()}
            table = torch.cat([self.x, self.gc_x])
        else:
            sub = fdata.assemble_subgraphs_torch(torch.from_numpy(ei).to(dev), N, co.assign, n, extra_node=extra_node)
            table = self.x
        self.sub_ptr = sub["ptr"].cpu().numpy()                       # union rows of every cluster subgraph
        self.gs_node = sub["node_id"]
        self.gs_core = sub["core"]
        self.gs_edge_index = sub["edge_index"]
        self.gs_x = table[self.gs_node].contiguous()
        R = int(self.sub_ptr[-1])
        sub_of_row = torch.repeat_interleave(torch.arange(n, device=dev), torch.from_numpy(np.diff(self.sub_ptr)).to(dev))
        n_core = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, sub_of_row, self.gs_core.long())
        pos_in_sub = torch.arange(R, device=dev) - torch.from_numpy(self.sub_ptr[:-1]).to(dev)[sub_of_row]
        self.gs_mask = (pos_in_sub < n_core[sub_of_row]) if reference_mask else self.gs_core.clone()
        self.gs_graph = self.gc_graph[sub_of_row]                     # graph of every union row
        self.gs_ptr = self.sub_ptr[self.cluster_ptr]                  # union rows of every graph

    # ---- batches: contiguous graph ranges ------------------------------------------------------------------
    def _flat(self, kind):
        """(row pointer per graph, x, mask | None, edge list grouped by graph, edge pointer per graph) of one view, cached."""
        cache = self.__dict__.setdefault("_flat_cache", {})
        if kind not in cache:
            dev = self.x.device
            if kind == "orig":
                ptr, x, mask, ei = self.node_ptr, self.x, None, self.edge_index
            elif kind == "gs":
                ptr, x, mask, ei = self.gs_ptr, self.gs_x, self.gs_mask, self.gs_edge_index
            else:
                ptr, x, mask, ei = self.cluster_ptr, self.gc_x, None, self.gc_edge_index
            ptr_t = torch.as_tensor(np.asarray(ptr), dtype=torch.int64, device=dev)
            eg = torch.searchsorted(ptr_t, ei[0].contiguous(), right=True) - 1       # graph of every edge (by its source row)
            order = torch.argsort(eg, stable=True)
            eptr = torch.searchsorted(eg[order].contiguous(), torch.arange(self.n_graphs + 1, device=dev))
            cache[kind] = (ptr_t, x, mask, ei[:, order].contiguous(), eptr)
        return cache[kind]

    def batch_ids(self, ids, kind):
        """The graphs `ids` (any order, no repeats needed) as one block-diagonal piece, like batch(): rows and edges are
        gathered with two index vectors built from the per-graph pointers -- no per-graph scan of the edge list, which is
        what GraphTrainer(reshuffle=True) pays for when it re-draws the batches every epoch (run.py:710)."""
        dev = self.x.device
        ptr, x, mask, ei, eptr = self._flat(kind)
        ids_t = torch.as_tensor(ids, dtype=torch.int64, device=dev)
        starts, lens = ptr[ids_t], ptr[ids_t + 1] - ptr[ids_t]
        new0 = torch.cumsum(lens, 0) - lens                                            # first new row of every graph
        shift = starts - new0
        total = int(lens.sum())
        rows = torch.repeat_interleave(shift, lens) + torch.arange(total, device=dev)
        es, el = eptr[ids_t], eptr[ids_t + 1] - eptr[ids_t]
        n_e = int(el.sum())
        eidx = torch.repeat_interleave(es - (torch.cumsum(el, 0) - el), el) + torch.arange(n_e, device=dev)
        e = (ei[:, eidx] - torch.repeat_interleave(shift, el)).contiguous()
        graph = torch.repeat_interleave(torch.arange(len(ids), device=dev), lens)
        if dev.type == "cuda" and total > 0:
            register(e, CSRGraph(e, total, mode="gcn"), "gcn")
        return dict(x=x[rows], edge_index=e, graph=graph, mask=None if mask is None else mask[rows], y=self.y[ids_t],
                    n_graphs=len(ids))

    def batch(self, g0, g1, kind):
        """Graphs g0:g1 as one block-diagonal piece: dict(x, edge_index, graph (0-based), mask, y, n_graphs) for
        kind 'gs' (subgraph union), 'gc' (coarse graphs) or 'orig' (the uncoarsened graphs: baselines).  The piece's CSR is
        built and registered once."""
        dev = self.x.device
        if kind == "orig":
            r0, r1 = int(self.node_ptr[g0]), int(self.node_ptr[g1])
            x, graph, mask, ei = self.x[r0:r1], self.node_graph[r0:r1] - g0, None, self.edge_index
            blocks = self.node_ptr[g0:g1 + 1] - r0
        elif kind == "gs":
            r0, r1 = int(self.gs_ptr[g0]), int(self.gs_ptr[g1])
            x, graph, mask, ei, ptr = self.gs_x[r0:r1], self.gs_graph[r0:r1] - g0, self.gs_mask[r0:r1], self.gs_edge_index, self.sub_ptr
            blocks = ptr[(ptr >= r0) & (ptr <= r1)] - r0
        else:
            r0, r1 = int(self.cluster_ptr[g0]), int(self.cluster_ptr[g1])
            x, graph, mask, ei = self.gc_x[r0:r1], self.gc_graph[r0:r1] - g0, None, self.gc_edge_index
            blocks = self.cluster_ptr[g0:g1 + 1] - r0
        sel = (ei[0] >= r0) & (ei[0] < r1)
        e = (ei[:, sel] - r0).contiguous()
        if dev.type == "cuda" and r1 > r0:
            register(e, CSRGraph(e, r1 - r0, mode="gcn", ptr=blocks), "gcn")
        return dict(x=x, edge_index=e, graph=graph, mask=mask, y=self.y[g0:g1], n_graphs=g1 - g0)


class PaddedBatchPlan:
    """Batches of ANY graphs of a GraphSet's subgraph view ('gs') or coarse view ('gc') assembled on the device into buffers of fixed capacity, so that one
    captured hipGraph serves every batch of every epoch of a reshuffling loader (run.py:710 DataLoader(shuffle=True);
    train.GraphTrainer(reshuffle=True, capture=True)).

    Built once per (set, batch size): the whole dataset's normalised CSR (rows, entries and row tiles of a graph contiguous and
    cut at graph boundaries), its pooled-row lists, the first layer's aggregated input A_hat x of EVERY row (ops.aggregated_input's
    product, formed once for the dataset: A_hat is block diagonal per graph, so a batch's A_hat x is a gather of rows), the truncated
    targets.  `assemble()` issues fitgnn_batch_offsets + fitgnn_batch_gather for the batch perm[step B .. + B) (step: a device
    counter the first kernel advances); `batch` is the static batch dict those launches fill, shaped like GraphSet.batch_ids':
    x / edge_index are stand-ins whose cached CSR graph, aggregated input and pool index are the plan's buffers.
    Capacities: mean + `sigmas` standard deviations of a batch's totals (rows, entries, tiles, pooled rows): GEMMs and SpMMs of the
    step run over R_cap rows, the surplus rows being empty; `fits(ids)` tells the host -- which knows every graph's sizes -- whether a
    batch fits (a batch that does not is run the eager way)."""

    def __init__(self, gset, batch_size, target_fn, kind="gs", sigmas=6.0):
        """kind 'gs': the subgraph union (pooled rows = the reference's row mask); 'gc': the coarse graphs (every row pooled)."""
        from . import _lib, ops

        dev = gset.x.device
        B = int(batch_size)
        assert dev.type == "cuda" and 1 <= B <= 1024 and kind in ("gs", "gc")
        G = gset.n_graphs
        self.gset, self.B, self.dev, self.kind = gset, B, dev, kind
        if kind == "gs":
            g_row, blocks, x, pooled_mask = np.asarray(gset.gs_ptr, dtype=np.int64), np.asarray(gset.sub_ptr, dtype=np.int64), gset.gs_x, gset.gs_mask
        else:
            g_row = np.asarray(gset.cluster_ptr, dtype=np.int64)
            blocks, x = g_row, gset.gc_x
            pooled_mask = torch.ones(int(g_row[-1]), dtype=torch.bool, device=dev)
        R = int(g_row[-1])
        whole = gset.batch(0, G, kind)                       # the dataset as ONE block-diagonal batch: its CSR, normalised
        self._whole = whole                                  # (the CSR cache is keyed on this edge tensor: keep it alive)
        g_all = ops_csr_for(whole["edge_index"], R)
        f, t = g_all.f, g_all.t
        if not (torch.equal(f.rowptr, t.rowptr) and torch.equal(f.col, t.col) and torch.equal(f.val, t.val)):
            raise ValueError("the union's normalised adjacency is not symmetric: its transpose cannot share the batch's CSR")
        rp_host = f.rowptr.cpu().numpy().astype(np.int64)
        g_nnz = rp_host[g_row]
        # row tiles: the usual packing of the diagonal blocks (cluster subgraphs / coarse graphs), cut at every graph boundary (a
        # tile's window is its own rows, and graph boundaries are block boundaries: both halves of a cut tile are valid tiles)
        from .csr import make_tiles, TILE_INTS
        tiles, g_tile = cut_tiles_at_graphs(make_tiles(blocks, g_all.window_rows), g_row, rp_host)
        pooled = pooled_mask.to(torch.uint8).contiguous()
        mem = torch.nonzero(pooled_mask).flatten().to(torch.int32).contiguous()           # ascending: grouped by graph
        mem_host = mem.cpu().numpy().astype(np.int64)
        g_mem = np.searchsorted(mem_host, g_row, side="left")
        # rank of a pooled row among its graph's pooled rows (the compact position of the row inside its graph)
        rank = np.zeros(R, dtype=np.int32)
        graph_of_mem = np.searchsorted(g_row, mem_host, side="right") - 1
        rank[mem_host] = (np.arange(len(mem_host)) - g_mem[graph_of_mem]).astype(np.int32)
        self.mem_rank = torch.from_numpy(rank).to(dev)
        self.sizes = np.stack([np.diff(g_row), np.diff(g_nnz), np.diff(g_tile), np.diff(g_mem)], 1).astype(np.int64)   # [G, 4]
        mean, std = self.sizes.mean(0), self.sizes.std(0)
        cap = np.ceil(B * mean + sigmas * np.sqrt(B) * std + 1).astype(np.int64)
        cap = np.minimum(cap, np.sort(self.sizes, 0)[::-1][:B].sum(0))                   # never beyond the true worst case
        self.R_cap = int((cap[0] + 63) // 64 * 64)
        self.E_cap = int(cap[1] + 64)
        self.T_cap = int(cap[2] + self.R_cap // 16 + 2)      # + the tiles that cover the rows past a batch's total
        self.M_cap = int(cap[3] + 64)
        self.caps = np.array([self.R_cap, self.E_cap, int(cap[2]), self.M_cap], dtype=np.int64)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)  # noqa: E731
        self.g_row, self.g_nnz, self.g_tile, self.g_mem = i32(g_row), i32(g_nnz), i32(g_tile), i32(g_mem)
        self.rowptr, self.col, self.val = f.rowptr, f.col, f.val
        self.tiles, self.mem, self.pooled = torch.from_numpy(tiles).to(dev), mem, pooled
        self.K = int(x.shape[1])
        self.ax = ops.aggregated_input(g_all, x).contiguous()                             # [R, K]
        self.tgt = target_fn(gset.y).contiguous()                                         # [G, 1]
        self.n_tgt = int(self.tgt.shape[1])
        # ---- the batch's buffers ----
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)                            # noqa: E731
        self.b_rowptr, self.b_col, self.b_val = z(self.R_cap + 1, torch.int32), z(self.E_cap, torch.int32), z(self.E_cap, torch.float32)
        self.b_tiles = torch.zeros((self.T_cap, TILE_INTS), dtype=torch.int32, device=dev)
        self.b_members, self.b_seg_off = z(self.M_cap, torch.int32), z(B + 1, torch.int32)
        self.b_members64, self.b_cseg, self.b_pos = z(self.M_cap, torch.int64), z(self.M_cap, torch.int32), z(self.R_cap, torch.int32)
        self.b_seg_of_row, self.b_inv_cnt = z(self.R_cap, torch.int32), z(B, torch.float32)
        self.b_ax = torch.zeros((self.R_cap, self.K), dtype=torch.float32, device=dev)
        self.b_tgt = torch.zeros((B, self.n_tgt), dtype=torch.float32, device=dev)
        self.off, self.gid = z(4 * (B + 1), torch.int32), z(B, torch.int32)
        self.step_idx = z(1, torch.int32)
        self.loss_slot, self.loss_sum = z(1, torch.float32), z(1, torch.float32)
        self.perm = None                                     # int64 [>= steps x B]: set_epoch()
        self._zero_rows = int(ops.ZERO_ROWS)
        self.batch = self._static_batch(g_all)
        self._L, self._lib = _lib.lib(), _lib

    def _static_batch(self, g_all):
        """The batch dict the model is stepped on: stand-in x / edge_index / index tensors whose cached objects are the plan's buffers."""
        from . import ops
        from .csr import CSRGraph, _Side, register

        dev, B = self.dev, self.B
        side = _Side(self.b_rowptr, self.b_col)
        side.val, side.tiles, side.n_tiles = self.b_val, self.b_tiles, self.T_cap
        g = object.__new__(CSRGraph)
        g.device, g.n, g.mode, g.nnz = dev, self.R_cap, "gcn", self.E_cap
        g.planned, g.gather, g.split_large, g.block_limit = False, False, False, None
        g.fold_ok, g.f, g.t = False, side, side               # symmetric normalised adjacency: the transpose is the matrix itself
        g.dinv, g.seg, g.range_seg, g.window_rows, g.ptr = None, None, None, g_all.window_rows, None
        x = torch.zeros((self.R_cap, self.K), dtype=torch.float32, device=dev)            # never read: the first layer runs on b_ax
        e = torch.zeros((2, 1), dtype=torch.int64, device=dev)
        register(e, g, "gcn")
        g._agg_input = (x, x._version, self.b_ax)
        pi = object.__new__(ops.PoolIndex)
        pi.n_seg, pi.sorted, pi.seg_off, pi.members = B, True, self.b_seg_off, self.b_members
        pi.n_rows, pi.seg_of_row, pi.inv_cnt = self.R_cap, self.b_seg_of_row, self.b_inv_cnt
        self.graph = g
        if self.kind == "gs":
            rows = self.b_members64                                                       # mask_idx: the pooled rows (filled per batch)
            batch = torch.zeros(self.M_cap, dtype=torch.int64, device=dev)                # stand-in for graph_of_masked
            # the pool over x[rows] ...
            cache = {(batch._version, B, (rows.data_ptr(), rows._version), self.R_cap): (pi, rows)}
            # ... and over the compact pooled rows (a last layer evaluated on them only, ops.FusedGCNLayerRows): members = 0 .. M_cap - 1
            pc = object.__new__(ops.PoolIndex)
            pc.n_seg, pc.sorted, pc.seg_off = B, True, self.b_seg_off
            pc.members = torch.arange(self.M_cap, dtype=torch.int32, device=dev)
            pc.n_rows, pc.seg_of_row, pc.inv_cnt = self.M_cap, self.b_cseg, self.b_inv_cnt
            cache[(batch._version, B, None, self.M_cap)] = (pc, None)
            batch._fitgnn_pool = cache
            g._compact_pos = (rows, rows._version, self.b_pos)                            # ops._compact_positions(g, rows)
            return dict(x=x, edge_index=e, mask=None, mask_idx=rows, graph_of_masked=batch, y=self.b_tgt, n_graphs=B)
        import types
        batch = torch.zeros(self.R_cap, dtype=torch.int64, device=dev)                    # stand-in for the PyG batch vector
        batch._fitgnn_pool = {(batch._version, B, None, self.R_cap): (pi, None)}
        return dict(x=x, edge_index=e, mask=None, y=self.b_tgt, n_graphs=B,
                    gc=types.SimpleNamespace(x=x, edge_index=e, batch=batch, num_graphs=B))

    def fits(self, ids):
        """ids [n, B] (host): which of the n batches fit the capacities."""
        tot = self.sizes[np.asarray(ids, dtype=np.int64)].sum(1)                           # [n, 4]
        return (tot <= self.caps[None, :]).all(1)

    def set_epoch(self, ids_flat):
        """Upload the epoch's graph order (host int64 array, whole batches) and rewind the step counter and the loss sum."""
        t = torch.from_numpy(np.ascontiguousarray(ids_flat, dtype=np.int64))
        if self.perm is None or self.perm.numel() < t.numel():
            self.perm = torch.zeros(max(t.numel(), 1), dtype=torch.int64, device=self.dev)
        self.perm[: t.numel()].copy_(t)
        self.step_idx.zero_(); self.loss_sum.zero_(); self.loss_slot.zero_()

    def assemble(self):
        """The two launches that fill the static batch with perm[step B .. + B) (capturable)."""
        L, lb = self._L, self._lib
        st = lb.stream_ptr(self.dev)
        d = lb.dptr
        lb.check(L.fitgnn_batch_offsets(d(self.perm), d(self.step_idx), self.B, d(self.g_row), d(self.g_nnz), d(self.g_tile), d(self.g_mem),
                                        d(self.off), d(self.gid), d(self.loss_slot), d(self.loss_sum), st), "fitgnn_batch_offsets")
        lb.check(L.fitgnn_batch_gather(self.B, d(self.off), d(self.gid), d(self.g_row), d(self.g_nnz), d(self.g_tile), d(self.g_mem),
                                       d(self.rowptr), d(self.col), d(self.val), d(self.tiles), d(self.mem), d(self.pooled), d(self.ax),
                                       self.ax.stride(0), d(self.tgt), self.n_tgt, self.K, self.R_cap, self.E_cap, self.T_cap, self.M_cap,
                                       d(self.b_rowptr), d(self.b_col), d(self.b_val), d(self.b_tiles), d(self.b_members), d(self.b_seg_off),
                                       d(self.b_seg_of_row), d(self.b_inv_cnt), d(self.b_ax), self.b_ax.stride(0), d(self.b_tgt), d(self.mem_rank),
                                       d(self.b_members64), d(self.b_cseg), d(self.b_pos), self._zero_rows, st),
                 "fitgnn_batch_gather")


def cut_tiles_at_graphs(base, g_row, rowptr):
    """Contiguous-window row tiles (csr.make_tiles: a partition of rows 0 .. R into runs of whole diagonal blocks, window == the tile's
    own rows) cut at every graph boundary g_row[1:-1], so that no tile spans two graphs and a graph's tiles can be moved with its rows.
    Graph boundaries are block boundaries: both halves of a cut tile are valid tiles.  Returns (tiles int32 [T, 8] in row order with
    nnz_begin / nnz_end from `rowptr`, g_tile int64 [G + 1]: graph g owns tiles g_tile[g] .. g_tile[g + 1])."""
    from .csr import TILE_INTS

    g_row = np.asarray(g_row, dtype=np.int64)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    R = int(g_row[-1])
    starts = np.union1d(np.asarray(base["row_begin"], dtype=np.int64), g_row[:-1])
    starts = starts[starts < R]
    ends = np.append(starts[1:], R)
    tiles = np.zeros((len(starts), TILE_INTS), dtype=np.int32)
    tiles[:, 0], tiles[:, 1], tiles[:, 2], tiles[:, 3] = starts, ends, starts, ends - starts
    tiles[:, 4], tiles[:, 5] = rowptr[starts], rowptr[ends]
    return tiles, np.searchsorted(starts, g_row, side="left")


def ops_csr_for(edge_index, n):
    from .csr import csr_for

    return csr_for(edge_index, n, "gcn")
