"""Coarsened-subgraph batches: assembly of the per-cluster subgraphs Gs and their static block-diagonal
union, device resident.

Reference behaviour restated (utils.py, node-level tasks):
  * one subgraph per cluster (utils.py:184-267): node set `value` = the cluster's nodes, plus
      --extra_node   : every 1-hop neighbour outside the cluster           (:235-239, nodes_2_neighbours :58-62)
      (neither flag) : nothing
    sorted ascending (:243); M = data.subgraph(value) = the induced subgraph, nodes relabelled by rank (:248);
  * per-subgraph train/val/test masks = the dataset masks on the cluster's own nodes, False on extra
    nodes (load_data_classification, utils.py:683-703);
  * batches: G_DataLoader(batch_size=128, shuffle=False) (run.py:336) -> block-diagonal union, static order.
The reference builds this with Python loops that rescan all E edges per node (utils.neighbour :52-56);
here it is a handful of vectorised sorts/searches (host numpy; SURVEY §8 f1 lists a device version as next).
"""
import numpy as np
import torch

from . import csr as _csr
from .csr import CSRGraph


def synthetic_graph(N, E, seed=0):
    """Connected preferential-attachment graph with exactly N nodes and E undirected edges (E >= 2N-3 uses
    m=2 attachments per node, the remainder are uniformly random extra edges).  Returns directed
    edge_index [2, 2E] (both directions, int64 numpy), sorted by (src, dst)."""
    rng = np.random.default_rng(seed)
    m = 2 if E >= 2 * N - 3 else 1
    targets = list(range(m))
    repeated = []
    src, dst = [], []
    for v in range(m, N):
        for t in set(targets):
            src.append(v)
            dst.append(t)
        repeated.extend(set(targets))
        repeated.extend([v] * m)
        idx = rng.integers(0, len(repeated), size=m)
        targets = [repeated[i] for i in idx]
    if E > 1_000_000:  # large stand-ins (S-products): the uniformly random remainder drawn and de-duplicated in bulk
        base = np.array([(min(a, b), max(a, b)) for a, b in zip(src, dst)], dtype=np.int64)
        keys = np.unique(base[:, 0] * N + base[:, 1])
        while len(keys) < E:
            need = E - len(keys)
            a = rng.integers(0, N, size=need + need // 8 + 16)
            b = rng.integers(0, N, size=len(a))
            ok = a != b
            lo, hi = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
            fresh = np.setdiff1d(np.unique(lo * N + hi), keys)
            keys = np.union1d(keys, rng.permutation(fresh)[:need])
        und = np.stack([keys // N, keys % N], axis=1)
        ei = np.concatenate([und.T, und.T[::-1]], axis=1)
        order = np.lexsort((ei[1], ei[0]))
        return np.ascontiguousarray(ei[:, order])   # (a column-indexed view is not row-major: collectives ship memory as it lies)
    und = {(min(a, b), max(a, b)) for a, b in zip(src, dst)}
    while len(und) < E:
        need = E - len(und)
        a = rng.integers(0, N, size=2 * need + 16)
        b = rng.integers(0, N, size=2 * need + 16)
        for x, y in zip(a, b):
            if x != y:
                und.add((min(x, y), max(x, y)))
                if len(und) >= E:
                    break
    und = np.array(sorted(und), dtype=np.int64)[:E]
    ei = np.concatenate([und.T, und.T[::-1]], axis=1)
    order = np.lexsort((ei[1], ei[0]))
    return np.ascontiguousarray(ei[:, order])


def assemble_subgraphs(edge_index, num_nodes, assign, n_clusters, extra_node=True):
    """All cluster subgraphs at once.

    edge_index: int64 [2, E] directed (both directions present), assign: int [N] cluster id per node.
    Returns dict with (everything numpy):
      ptr        int64 [n_clusters+1]  rows of subgraph c are ptr[c]:ptr[c+1] of the union
      node_id    int64 [R]             original node of each union row (sorted ascending inside a subgraph)
      core       bool  [R]             True for the cluster's own nodes, False for extra nodes
      edge_index int64 [2, E']         union-row indices, block diagonal
    """
    src, dst = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    assign = np.asarray(assign, dtype=np.int64)
    N = int(num_nodes)
    # membership pairs (cluster, node): core pairs, plus for --extra_node the outside end of every cut edge
    pc = [assign, ]
    pn = [np.arange(N, dtype=np.int64), ]
    if extra_node:
        cut = assign[src] != assign[dst]
        pc.append(assign[src[cut]])
        pn.append(dst[cut])
    key = np.unique(np.concatenate(pc) * N + np.concatenate(pn))  # sorted by (cluster, node)
    mem_c, mem_n = key // N, key % N
    ptr = np.zeros(n_clusters + 1, dtype=np.int64)
    np.cumsum(np.bincount(mem_c, minlength=n_clusters), out=ptr[1:])
    core = assign[mem_n] == mem_c
    # induced edges: for every member row (c, x) and every neighbour y of x, keep it when (c, y) is a member
    order = np.lexsort((dst, src))
    s_src, s_dst = src[order], dst[order]
    adj_ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(s_src, minlength=N), out=adj_ptr[1:])
    deg = adj_ptr[mem_n + 1] - adj_ptr[mem_n]
    rows = np.repeat(np.arange(len(mem_n), dtype=np.int64), deg)          # union row of x
    starts = np.repeat(adj_ptr[mem_n], deg)
    within = np.arange(len(rows), dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)
    nbr = s_dst[starts + within]                                            # y
    want = mem_c[rows] * N + nbr
    pos = np.searchsorted(key, want)
    pos[pos >= len(key)] = len(key) - 1
    hit = key[pos] == want
    e_src, e_dst = rows[hit], pos[hit]                                      # x -> y inside the union
    return dict(ptr=ptr, node_id=mem_n, core=core, edge_index=np.stack([e_src, e_dst]))


def cluster_weights_torch(edge_index, num_nodes, assign, n_clusters, extra_node=True):
    """A cheap stand-in for nnz' per cluster subgraph, from the partition alone (no assembly): rows (own + distinct extra nodes) +
    directed edges with both ends among the own nodes + both directions of every (own node, extra node) edge.  What it leaves
    out are the edges BETWEEN extra nodes of a subgraph (~10 % of nnz' on S-products), so it is a lower bound that ranks
    clusters like nnz' does: data-parallel ranks shard on it BEFORE assembling (every rank computes the same numbers from the
    graph and the partition it already holds) and then assemble only their own clusters.  int64 numpy [n_clusters]."""
    dev = edge_index.device
    src, dst = edge_index[0].long(), edge_index[1].long()
    assign = torch.as_tensor(assign, device=dev).long()
    N, n = int(num_nodes), int(n_clusters)
    w = torch.bincount(assign, minlength=n)                                  # own nodes
    cs, cd = assign[src], assign[dst]
    inner = cs == cd
    w = w + torch.bincount(cs[inner], minlength=n)                           # own -- own, both directions are listed
    if extra_node:
        cut = ~inner
        w = w + 2 * torch.bincount(cs[cut], minlength=n)                     # own -> extra and back
        pairs = torch.unique(cs[cut] * N + dst[cut])                         # distinct (cluster, extra node)
        w = w + torch.bincount(pairs // N, minlength=n)
    return w.cpu().numpy()


def assemble_subgraphs_torch(edge_index, num_nodes, assign, n_clusters, extra_node=True, chunk_rows=1 << 20, layout="sorted",
                             clusters=None):
    """assemble_subgraphs with torch tensor ops on the device of `edge_index` (the reference's
    neighbour() scans all E edges per node, utils.py:52-56; here: one sort of the membership keys, one CSR gather and a
    binary search per (member, neighbour) pair, in chunks of `chunk_rows` members to bound memory).
    Same dict as assemble_subgraphs, values are int64 / bool tensors on that device (ptr too).

    layout: the order of a subgraph's rows in the union -- an internal choice: results per node do not depend on it.
      "sorted"  ascending node id, the reference's order (utils.py:243).
      "star"    star by star: every own node of the cluster (ascending) followed by the extra nodes whose lowest-numbered
                own neighbour it is (ascending).  An --extra_node subgraph is a bundle of stars (the own nodes are the
                centres: every extra node is there because it neighbours one); laid out like this each star is a run of
                consecutive rows that references little outside itself, which is what the whole-subgraph SpMM kernel wants
                (csrc/spmm.hip: every operand row read once).  The dict then also carries `seg_start` (bool per row: first
                row of a star).
    clusters (ascending cluster ids, optional): assemble ONLY these clusters' subgraphs -- the result is what
      select_clusters(assemble_subgraphs_torch(...), clusters) returns (clusters renumbered 0..len-1 in that order) without ever
      building the others: a data-parallel rank's shard."""
    dev = edge_index.device
    src, dst = edge_index[0].long(), edge_index[1].long()
    assign = torch.as_tensor(assign, device=dev).long()
    N, n = int(num_nodes), int(n_clusters)
    nodes = torch.arange(N, device=dev)
    if clusters is not None:
        # renumber: selected cluster -> its rank in `clusters`, every other cluster -> -1; edges and nodes that start in an
        # unselected cluster take no part in the membership list
        sel = torch.as_tensor(np.asarray(clusters, dtype=np.int64), device=dev)
        new_id = torch.full((n,), -1, dtype=torch.int64, device=dev)
        new_id[sel] = torch.arange(int(sel.numel()), device=dev)
        assign = new_id[assign]
        n = int(sel.numel())
        nodes = nodes[assign >= 0]
    pc, pn = [assign[nodes]], [nodes]
    if extra_node:
        # cut edges that START in a (selected) cluster; the full edge list stays: an extra node's own adjacency is needed below
        cut = (assign[src] >= 0) & (assign[src] != assign[dst])
        pc.append(assign[src[cut]])
        pn.append(dst[cut])
    key = torch.unique(torch.cat(pc) * N + torch.cat(pn))            # sorted by (cluster, node)
    mem_c, mem_n = key // N, key % N
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(mem_c, minlength=n), 0)
    core = assign[mem_n] == mem_c
    inv = None
    seg_start = None
    if layout == "star":
        R0 = int(key.numel())
        hub = mem_n.clone()                                           # own node: its own star
        if extra_node and bool(cut.any()):
            pos_x = torch.searchsorted(key, assign[src[cut]] * N + dst[cut])       # row of (cluster of src, dst): an extra node
            first = torch.full((R0,), N, dtype=torch.int64, device=dev).scatter_reduce(0, pos_x, src[cut], reduce="amin")
            hub = torch.where(core, mem_n, first)
        star_key = (mem_c * N + hub) * 2 + (~core).long()              # rows are already in (cluster, node) order: stable sort
        perm = torch.argsort(star_key, stable=True)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(R0, device=dev)
        mem_c, mem_n, core, hub = mem_c[perm], mem_n[perm], core[perm], hub[perm]
        seg_start = torch.ones(R0, dtype=torch.bool, device=dev)
        seg_start[1:] = (mem_c[1:] != mem_c[:-1]) | (hub[1:] != hub[:-1])
    elif layout != "sorted":
        raise ValueError(f"layout {layout!r}: 'sorted' or 'star'")
    order = torch.argsort(src * N + dst)
    s_dst = dst[order]
    adj_ptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    adj_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=N), 0)
    R = int(key.numel())
    if dev.type == "cuda":
        # the induced edges by the library's kernels (csrc/assemble.hip): every member row walks its node's adjacency list once and
        # looks the neighbours up inside its own cluster's member run -- no (row, neighbour) temporaries, a 7-step search in L1
        from . import _lib
        L, st = _lib.lib(), _lib.stream_ptr(dev)
        key_node = (key % N).contiguous()
        s_dst, adj_ptr = s_dst.contiguous(), adj_ptr.contiguous()
        mem_n, mem_c, ptr_c = mem_n.contiguous(), mem_c.contiguous(), ptr.contiguous()
        cnt = torch.empty(R, dtype=torch.int32, device=dev)
        _lib.check(L.fitgnn_induced_edges_count(_lib.dptr(adj_ptr), _lib.dptr(s_dst), _lib.dptr(mem_n), _lib.dptr(mem_c), _lib.dptr(ptr_c),
                                                _lib.dptr(key_node), R, _lib.dptr(cnt), st), "fitgnn_induced_edges_count")
        off = torch.zeros(R + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(cnt, 0)
        n_e = int(off[-1])
        e = torch.empty((2, n_e), dtype=torch.int64, device=dev)
        _lib.check(L.fitgnn_induced_edges_fill(_lib.dptr(adj_ptr), _lib.dptr(s_dst), _lib.dptr(mem_n), _lib.dptr(mem_c), _lib.dptr(ptr_c),
                                               _lib.dptr(key_node), _lib.dptr(None if inv is None else inv.contiguous()), R, _lib.dptr(off),
                                               _lib.dptr(e[0]), _lib.dptr(e[1]), st), "fitgnn_induced_edges_fill")
        out = dict(ptr=ptr, node_id=mem_n, core=core, edge_index=e)
        if seg_start is not None:
            out["seg_start"] = seg_start
        return out
    deg = adj_ptr[mem_n + 1] - adj_ptr[mem_n]
    es, ed = [], []
    for r0 in range(0, R, chunk_rows):
        r1 = min(R, r0 + chunk_rows)
        d = deg[r0:r1]
        rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), d)
        first = torch.cumsum(d, 0) - d
        within = torch.arange(int(rows.numel()), device=dev) - torch.repeat_interleave(first, d)
        nbr = s_dst[torch.repeat_interleave(adj_ptr[mem_n[r0:r1]], d) + within]
        want = mem_c[rows] * N + nbr
        pos = torch.searchsorted(key, want).clamp_(max=R - 1)
        hit = key[pos] == want
        es.append(rows[hit]); ed.append(pos[hit] if inv is None else inv[pos[hit]])   # key positions -> rows of the layout
    out = dict(ptr=ptr, node_id=mem_n, core=core, edge_index=torch.stack([torch.cat(es), torch.cat(ed)]))
    if seg_start is not None:
        out["seg_start"] = seg_start
    return out


def assemble_subgraphs_cluster(edge_index, num_nodes, assign, n_clusters, coarse_adj):
    """--cluster_node subgraphs (utils.py:190-232, :252-259), all clusters at once.

    Cluster c's subgraph = its own nodes (ascending) followed by one NEW node per neighbouring cluster d, in the
    reference's order of first appearance (members ascending, and per member its connected clusters ascending);
    a new node carries the pooled features (C.X)[d], label 0, mask False.  Edges: the induced edges of the own nodes,
    member <-> new node for every (member, neighbouring cluster) pair, and new(d1) <-> new(d2) when the coarse graph
    links d1 and d2 (`adj[d1, d2] or adj[d2, d1]`).
    coarse_adj: scipy sparse [n_clusters x n_clusters], non-zero = coarse edge (Gc.A of every component, global ids).
    Returns the dict of assemble_subgraphs; node_id >= num_nodes encodes the new node of cluster node_id - num_nodes
    (rows of a feature table [X ; C.X])."""
    import scipy.sparse as sp

    src, dst = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    assign = np.asarray(assign, dtype=np.int64)
    N, n = int(num_nodes), int(n_clusters)
    cut = assign[src] != assign[dst]
    cx, cd = src[cut], assign[dst[cut]]                     # member x (in cluster assign[x]) sees cluster d
    cc = assign[cx]
    # distinct (c, d) pairs = new nodes; order inside c by (first member that sees d, d)
    pair_key, inv = np.unique(cc * n + cd, return_inverse=True)
    first_x = np.full(len(pair_key), N, dtype=np.int64)
    np.minimum.at(first_x, inv, cx)
    pc, pd = pair_key // n, pair_key % n
    order = np.lexsort((pd, first_x, pc))
    pc, pd, pair_sorted = pc[order], pd[order], pair_key[order]
    n_new = np.bincount(pc, minlength=n)
    n_core = np.bincount(assign, minlength=n)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(n_core + n_new, out=ptr[1:])
    # union rows: own nodes first (ascending id), then the new nodes in the order above
    core_order = np.lexsort((np.arange(N), assign))
    core_start = np.cumsum(n_core) - n_core
    row_of_node = np.empty(N, dtype=np.int64)
    row_of_node[core_order] = ptr[assign[core_order]] + (np.arange(N) - core_start[assign[core_order]])
    new_start = np.cumsum(n_new) - n_new
    new_row = ptr[pc] + n_core[pc] + (np.arange(len(pc)) - new_start[pc])
    R = int(ptr[-1])
    node_id = np.empty(R, dtype=np.int64)
    node_id[row_of_node] = np.arange(N)
    node_id[new_row] = N + pd
    core = np.zeros(R, dtype=bool)
    core[row_of_node] = True
    # lookup (c, d) -> new row
    look = np.argsort(pair_sorted, kind="stable")
    keys_sorted, rows_sorted = pair_sorted[look], new_row[look]
    def new_row_of(c, d):
        pos = np.searchsorted(keys_sorted, c * n + d)
        pos = np.minimum(pos, len(keys_sorted) - 1) if len(keys_sorted) else pos
        ok = (keys_sorted[pos] == c * n + d) if len(keys_sorted) else np.zeros(len(c), dtype=bool)
        return rows_sorted[pos] if len(keys_sorted) else pos, ok
    es, ed = [], []
    inn = ~cut
    es.append(row_of_node[src[inn]]); ed.append(row_of_node[dst[inn]])                 # induced edges of the own nodes
    xd = np.unique(cx * n + cd)                                                        # (member, neighbouring cluster)
    mx, md = xd // n, xd % n
    r_new, ok = new_row_of(assign[mx], md)
    assert bool(np.all(ok))
    es += [row_of_node[mx], r_new]; ed += [r_new, row_of_node[mx]]
    S = sp.csr_matrix(coarse_adj)
    S = ((S + S.T) != 0).tocsr()
    S.setdiag(False); S.eliminate_zeros()
    if len(pc):                                                                        # new(d1) - new(d2), d2 a coarse neighbour of d1
        deg = np.diff(S.indptr)[pd]
        rep = np.repeat(np.arange(len(pc)), deg)
        within = np.arange(len(rep)) - np.repeat(np.cumsum(deg) - deg, deg)
        d2 = S.indices[S.indptr[pd][rep] + within].astype(np.int64)
        r2, ok2 = new_row_of(pc[rep], d2)
        es.append(new_row[rep][ok2]); ed.append(r2[ok2])
    return dict(ptr=ptr, node_id=node_id, core=core,
                edge_index=np.stack([np.concatenate(es), np.concatenate(ed)]).astype(np.int64))


class SubgraphBatch:
    """Device-resident block-diagonal union of cluster subgraphs (one static 'batch of batches').

    The reference trains in GD mode by forwarding every 128-subgraph batch and summing ONE loss over all of
    them (run.py:184-204); subgraphs share no edges, so one union pass is the same arithmetic."""

    def __init__(self, sub, X, y, train_mask, device="cuda", lds_rows=None, dedup=True, float_targets=False):
        """dedup: keep ONE copy of every original node's features on the device (`x_table`, `row_index`) next to the
        materialised union rows `x`; models that accept `x_index` then run their first layer on the table."""
        dev = torch.device(device)
        as_t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(a)).to(dev)  # noqa: E731
        self.ptr = sub["ptr"].cpu().numpy() if torch.is_tensor(sub["ptr"]) else sub["ptr"]
        self.n_rows = int(self.ptr[-1])
        self.node_id = as_t(sub["node_id"])
        self.core = as_t(sub["core"])
        self.edge_index = as_t(sub["edge_index"])
        X = X if torch.is_tensor(X) else torch.from_numpy(np.asarray(X))
        Xd = X.to(dev).float()
        self.x = Xd[self.node_id].contiguous()
        self.x_table, self.row_index = None, None
        if dedup and dev.type == "cuda":
            from .ops import RowIndex
            self.x_table = Xd.contiguous()
            self.row_index = RowIndex(self.node_id, Xd.shape[0])
        y = y if torch.is_tensor(y) else torch.from_numpy(np.asarray(y))
        self.y = y.to(dev)[self.node_id].float() if float_targets else y.to(dev)[self.node_id].long()
        tm = train_mask if torch.is_tensor(train_mask) else torch.from_numpy(np.asarray(train_mask))
        self.train_mask = tm.to(dev)[self.node_id] & self.core          # utils.py:695-698
        self.train_idx = torch.nonzero(self.train_mask).flatten()
        self.graph = None
        # row runs the SpMM kernels treat as units: the stars of a star-by-star layout, else the subgraphs themselves
        self.seg_ptr = self.ptr
        if sub.get("seg_start") is not None:
            st = sub["seg_start"]
            st = st.cpu().numpy() if torch.is_tensor(st) else np.asarray(st)
            self.seg_ptr = np.concatenate([np.nonzero(st)[0], [self.n_rows]]).astype(np.int64)
        if dev.type == "cuda":
            self.graph = CSRGraph(self.edge_index, self.n_rows, mode="gcn", ptr=self.seg_ptr, lds_rows=lds_rows)
            _csr.register(self.edge_index, self.graph, "gcn")  # model(x, edge_index) finds it by identity
        self.nnz = int(self.edge_index.shape[1]) + self.n_rows           # nnz' = directed edges + self loops

    def register_mode(self, mode):
        """Pre-build (and register for `model(x, edge_index)` to find) this union's CSR in another layer's mode -- 'gat', 'sum', 'mean' --
        with the SAME row runs as the GCN one (the stars of a star-by-star layout), instead of letting the layer detect the diagonal
        blocks on its first call."""
        g = CSRGraph(self.edge_index, self.n_rows, mode=mode, ptr=self.seg_ptr)
        _csr.register(self.edge_index, g, mode)
        return g

    def slice_batches(self, batch_size=128):
        """(row_begin, row_end) of the reference's loader batches (run.py:336: 128 subgraphs each)."""
        c = len(self.ptr) - 1
        return [(int(self.ptr[b]), int(self.ptr[min(b + batch_size, c)])) for b in range(0, c, batch_size)]


def shard_clusters(ptr, nnz_per_cluster, world_size):
    """Static longest-processing-time assignment of whole subgraphs to ranks, balancing nnz' (SURVEY §8e)."""
    import heapq

    nnz = np.asarray(nnz_per_cluster, dtype=np.int64)
    order = np.argsort(-nnz, kind="stable")
    owner = np.zeros(len(order), dtype=np.int64)
    heap = [(0, r) for r in range(world_size)]   # (load, rank): ties go to the lowest rank, as argmin would
    for c in order:
        load, r = heapq.heappop(heap)
        owner[c] = r
        heapq.heappush(heap, (load + int(nnz[c]), r))
    return owner


def cluster_nnz(sub):
    """nnz' of every cluster subgraph of an assemble_subgraphs* result: its directed edges + one self loop per row (the
    non-zeros of A_hat a SpMM over the subgraph consumes; the unit shard_clusters balances)."""
    ptr = sub["ptr"]
    if torch.is_tensor(ptr):
        n = int(ptr.numel()) - 1
        src = sub["edge_index"][0]
        owner = torch.searchsorted(ptr, src, right=True) - 1
        return (torch.bincount(owner, minlength=n) + (ptr[1:] - ptr[:-1])).cpu().numpy()
    n = len(ptr) - 1
    owner = np.searchsorted(ptr, sub["edge_index"][0], side="right") - 1
    return np.bincount(owner, minlength=n) + np.diff(ptr)


def select_clusters(sub, clusters):
    """The sub-union holding only `clusters` (ascending cluster ids) of an assemble_subgraphs* result, rows renumbered:
    a rank's shard of ONE union (whole subgraphs are the unit; no edge crosses subgraphs, utils.py:248).
    Works on the NumPy and on the torch (device) form; returns the same dict layout."""
    if torch.is_tensor(sub["ptr"]):
        dev = sub["ptr"].device
        ptr = sub["ptr"].long()
        c = torch.as_tensor(np.asarray(clusters, dtype=np.int64), device=dev)
        size = ptr[c + 1] - ptr[c]
        new_ptr = torch.zeros(int(c.numel()) + 1, dtype=torch.int64, device=dev)
        new_ptr[1:] = torch.cumsum(size, 0)
        R = int(new_ptr[-1])
        rows = torch.repeat_interleave(ptr[c] - new_ptr[:-1], size) + torch.arange(R, device=dev)   # old row of every new row
        new_of_old = torch.full((int(ptr[-1]),), -1, dtype=torch.int64, device=dev)
        new_of_old[rows] = torch.arange(R, device=dev)
        e = sub["edge_index"]
        keep = new_of_old[e[0]] >= 0
        e2 = torch.stack([new_of_old[e[0][keep]], new_of_old[e[1][keep]]])
        out = dict(ptr=new_ptr, node_id=sub["node_id"][rows], core=sub["core"][rows], edge_index=e2)
        if "seg_start" in sub:
            out["seg_start"] = sub["seg_start"][rows]
        return out
    ptr = np.asarray(sub["ptr"], dtype=np.int64)
    c = np.asarray(clusters, dtype=np.int64)
    size = ptr[c + 1] - ptr[c]
    new_ptr = np.zeros(len(c) + 1, dtype=np.int64)
    np.cumsum(size, out=new_ptr[1:])
    R = int(new_ptr[-1])
    rows = np.repeat(ptr[c] - new_ptr[:-1], size) + np.arange(R, dtype=np.int64)
    new_of_old = np.full(int(ptr[-1]), -1, dtype=np.int64)
    new_of_old[rows] = np.arange(R, dtype=np.int64)
    e = np.asarray(sub["edge_index"])
    keep = new_of_old[e[0]] >= 0
    return dict(ptr=new_ptr, node_id=np.asarray(sub["node_id"])[rows], core=np.asarray(sub["core"])[rows],
                edge_index=np.stack([new_of_old[e[0][keep]], new_of_old[e[1][keep]]]))
