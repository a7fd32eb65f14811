"""CPU tier: host-side index plumbing (CSR build, block detection, tile packing, graph object)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from fitgnn_amd.coarsening import Graph
from fitgnn_amd.csr import CSRGraph, block_boundaries, make_tiles


def random_block_graph(sizes, seed=0, p=0.4):
    rng = np.random.default_rng(seed)
    src, dst, off = [], [], 0
    for s in sizes:
        for i in range(s):
            for j in range(i + 1, s):
                if rng.random() < p:
                    src += [off + i, off + j]
                    dst += [off + j, off + i]
        off += s
    return torch.tensor([src, dst], dtype=torch.long), off


def test_csr_and_transpose_match_scipy():
    ei, n = random_block_graph([5, 1, 9, 3, 30], seed=1)
    g = CSRGraph(ei, n, mode="gcn")
    A = sp.coo_matrix((np.ones(ei.shape[1]), (ei[1].numpy(), ei[0].numpy())), shape=(n, n)).tocsr()
    A = (A + sp.eye(n)).tocsr()
    A.sort_indices()
    assert np.array_equal(g.rowptr.numpy(), A.indptr) and np.array_equal(g.col.numpy(), A.indices)
    At = A.T.tocsr()
    At.sort_indices()
    assert np.array_equal(g.rowptr_t.numpy(), At.indptr) and np.array_equal(g.col_t.numpy(), At.indices)
    # perm_t carries forward values to the transposed slots
    vals = torch.arange(g.nnz, dtype=torch.float32)
    dense = torch.zeros(n, n)
    rows = torch.repeat_interleave(torch.arange(n), (g.rowptr[1:] - g.rowptr[:-1]).long())
    dense[rows, g.col.long()] = vals
    rows_t = torch.repeat_interleave(torch.arange(n), (g.rowptr_t[1:] - g.rowptr_t[:-1]).long())
    assert torch.equal(dense.t()[rows_t, g.col_t.long()], vals[g._perm_t])


def test_self_loops_replaced_not_duplicated():
    ei = torch.tensor([[0, 0, 1, 2, 2], [0, 1, 0, 2, 2]])
    g = CSRGraph(ei, 3, mode="gcn")
    assert g.rowptr.tolist() == [0, 2, 4, 5] and g.col.tolist() == [0, 1, 0, 1, 2]


def test_block_boundaries_and_tiles():
    sizes = [5, 1, 9, 3, 30, 2, 2, 70, 4]
    ei, n = random_block_graph(sizes, seed=2, p=1.0)
    g = CSRGraph(ei, n, mode="gcn")
    ptr = block_boundaries(g.rowptr, g.col, n).numpy()
    assert ptr.tolist() == np.concatenate([[0], np.cumsum(sizes)]).tolist()
    tiles = make_tiles(ptr, 64)
    # contiguous cover, each tile <= 64 rows, window == rows, no block split except the 70-node one
    assert tiles["row_begin"][0] == 0 and tiles["row_end"][-1] == n
    assert np.all(tiles["row_begin"][1:] == tiles["row_end"][:-1])
    assert np.all(tiles["row_end"] - tiles["row_begin"] <= 64)
    assert np.all(tiles["win_begin"] == tiles["row_begin"]) and np.all(tiles["win_rows"] == tiles["row_end"] - tiles["row_begin"])
    cuts = set(tiles["row_begin"].tolist())
    big0 = int(np.cumsum(sizes)[6])
    assert cuts - set(ptr.tolist()) == {big0 + 64}


def test_tiles_of_unstructured_matrix_cover_rows():
    tiles = make_tiles(np.array([0, 1000]), 64)
    assert len(tiles) == 16 and tiles["row_end"][-1] == 1000


def test_empty_graph():
    g = CSRGraph(torch.zeros((2, 0), dtype=torch.long), 0, mode="sum")
    assert g.nnz == 0 and g.rowptr.tolist() == [0]
    assert make_tiles(np.array([0]), 64).shape == (0,)


def test_graph_object_matches_pygsp_contract():
    W = sp.csr_matrix(np.array([[0, 2.0, 0], [2.0, 0, 1.0], [0, 1.0, 0]]))
    G = Graph(W)
    assert G.N == 3 and np.allclose(G.dw, [2, 3, 1])
    assert np.allclose(G.L.toarray(), np.diag([2, 3, 1]) - W.toarray())
    assert G.A.dtype == bool and G.A.nnz == 4
    comps = Graph(sp.block_diag([W, sp.csr_matrix((1, 1)), W[:2, :2]])).extract_components()
    assert [len(c.info["orig_idx"]) for c in comps] == [3, 1, 2]


def test_subgraph_assembly_matches_bruteforce():
    """assemble_subgraphs == the per-cluster construction of utils.py:186-267 (--extra_node) done naively."""
    from fitgnn_amd.data import assemble_subgraphs, synthetic_graph

    ei = synthetic_graph(300, 700, seed=3)
    assert ei.shape == (2, 1400)
    rng = np.random.default_rng(0)
    n_c = 90
    assign = rng.integers(0, n_c, size=300)
    assign[:n_c] = np.arange(n_c)
    for extra in (True, False):
        sub = assemble_subgraphs(ei, 300, assign, n_c, extra_node=extra)
        E = set(zip(ei[0].tolist(), ei[1].tolist()))
        adj = {}
        for a, b in E:
            adj.setdefault(a, set()).add(b)
        got_edges = set(zip(sub["edge_index"][0].tolist(), sub["edge_index"][1].tolist()))
        want_edges = set()
        for c in range(n_c):
            core = set(np.nonzero(assign == c)[0].tolist())
            value = set(core)
            if extra:
                for u in core:
                    value |= adj.get(u, set())
            value = sorted(value)
            r0, r1 = int(sub["ptr"][c]), int(sub["ptr"][c + 1])
            assert sub["node_id"][r0:r1].tolist() == value
            assert sub["core"][r0:r1].tolist() == [v in core for v in value]
            loc = {v: r0 + i for i, v in enumerate(value)}
            for u in value:
                for v in adj.get(u, ()):
                    if v in loc:
                        want_edges.add((loc[u], loc[v]))
        assert got_edges == want_edges


def test_tile_planner_invariants():
    """fitgnn_plan_tiles_host: contiguous row cover, row/window caps, lcol consistent with the windows."""
    from fitgnn_amd.csr import plan_tiles

    rng = np.random.default_rng(1)
    sizes = [3, 40, 1, 150, 9, 9, 2, 70, 5] * 3
    ei, n = random_block_graph(sizes, seed=5, p=0.15)
    # add hubs: node 0 of each big block linked to everyone in the block
    off, src, dst = 0, [], []
    for s_ in sizes:
        if s_ > 30:
            for j in range(1, s_):
                src += [off, off + j]; dst += [off + j, off]
        off += s_
    ei = torch.cat([ei, torch.tensor([src, dst], dtype=torch.long)], 1)
    g = CSRGraph(ei, n, mode="gcn")
    bptr = np.concatenate([[0], np.cumsum(sizes)])
    for max_rows, max_win, bp in ((16, 16, None), (8, 24, None), (32, 32, None), (4, 4, None), (16, 16, bptr), (24, 24, bptr)):
        tiles, win, lcol = plan_tiles(g.rowptr, g.col, n, max_rows, max_win, block_ptr=bp)
        if bp is not None:  # blocks that fit are never split
            cuts = set(tiles[:, 0].tolist())
            for b0, b1 in zip(bp[:-1], bp[1:]):
                if b1 - b0 <= min(max_rows, max_win):
                    assert not any(b0 < c < b1 for c in cuts)
        rp, col = g.rowptr.numpy(), g.col.numpy()
        assert tiles[0, 0] == 0 and tiles[-1, 1] == n and np.all(tiles[1:, 0] == tiles[:-1, 1])
        assert np.all(tiles[:, 1] - tiles[:, 0] <= max_rows) and np.all(tiles[:, 1] > tiles[:, 0])
        assert np.all(tiles[:, 3] <= max_win)
        assert np.all(tiles[:, 4] == rp[tiles[:, 0]]) and np.all(tiles[:, 5] == rp[tiles[:, 1]])
        hits = 0
        for t in tiles:
            wcols = win[t[2]: t[2] + t[3]] if t[6] else np.arange(t[2], t[2] + t[3])
            assert np.all(np.diff(wcols) > 0)
            for e in range(t[4], t[5]):
                if lcol[e] >= 0:
                    assert wcols[lcol[e]] == col[e]
                    hits += 1
                else:
                    assert -(lcol[e] + 1) == col[e]
        if max_win >= 16:
            assert hits / len(col) > 0.6  # dense random blocks cannot fit; star-shaped ones do (GPU test)


def test_xcd_tile_layout_is_balanced_and_complete():
    from fitgnn_amd.csr import TILE_INTS, arrange_tiles_for_xcds

    rng = np.random.default_rng(0)
    T = 1003
    rows = rng.integers(1, 17, size=T)
    rows[:100] = 16  # heavy head, light tail
    rb = np.concatenate([[0], np.cumsum(rows)])
    t = np.zeros((T, TILE_INTS), dtype=np.int32)
    t[:, 0], t[:, 1], t[:, 2], t[:, 3] = rb[:-1], rb[1:], rb[:-1], rows
    out = arrange_tiles_for_xcds(t)
    assert out.shape[0] % 8 == 0
    live = out[out[:, 1] > out[:, 0]]
    assert len(live) == T and sorted(live[:, 0].tolist()) == t[:, 0].tolist()
    per_xcd = [int((out[k::8, 1] - out[k::8, 0]).sum()) for k in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 2 * 16 + 16
    for k in range(8):  # each XCD walks a contiguous ascending range
        seg = out[k::8]
        seg = seg[seg[:, 1] > seg[:, 0]]
        assert np.all(seg[1:, 0] == seg[:-1, 1])


def test_cluster_node_assembly_matches_reference_construction():
    """assemble_subgraphs_cluster == the loop of utils.py:190-232 restated literally (new-node order, edges)."""
    import scipy.sparse as sp

    from fitgnn_amd.data import assemble_subgraphs_cluster, synthetic_graph

    N, n_c = 240, 60
    ei = synthetic_graph(N, 520, seed=5)
    rng = np.random.default_rng(2)
    assign = rng.integers(0, n_c, size=N)
    assign[:n_c] = np.arange(n_c)
    # coarse adjacency = lift of the edges (what Gc.A holds), upper triangle only to exercise the `or adj[j, i]` branch
    cu, cv = assign[ei[0]], assign[ei[1]]
    k = cu < cv
    adj = sp.csr_matrix((np.ones(int(k.sum())), (cu[k], cv[k])), shape=(n_c, n_c))
    adj.data[:] = 1
    sub = assemble_subgraphs_cluster(ei, N, assign, n_c, adj)
    nbrs = {}
    for a, b in zip(ei[0].tolist(), ei[1].tolist()):
        nbrs.setdefault(a, []).append(b)
    A = adj.toarray() > 0
    got = set(zip(sub["edge_index"][0].tolist(), sub["edge_index"][1].tolist()))
    assert len(got) == sub["edge_index"].shape[1]  # no duplicate edges
    want = set()
    for c in range(n_c):
        value = np.sort(np.nonzero(assign == c)[0])
        r0 = int(sub["ptr"][c])
        loc = {int(v): r0 + i for i, v in enumerate(value)}
        num_nodes = len(value)
        new_of = {}                                   # meta_node_2_new_node, insertion ordered
        for node in value.tolist():
            outside = [v for v in nbrs.get(node, []) if v not in loc]
            for cl in np.unique(assign[outside]).tolist() if outside else []:
                if cl not in new_of:
                    new_of[cl] = r0 + num_nodes
                    num_nodes += 1
                want.add((loc[node], new_of[cl])); want.add((new_of[cl], loc[node]))
        keys = list(new_of)
        for i in range(len(keys) - 1):
            for j in range(i + 1, len(keys)):
                if A[keys[i], keys[j]] or A[keys[j], keys[i]]:
                    want.add((new_of[keys[i]], new_of[keys[j]])); want.add((new_of[keys[j]], new_of[keys[i]]))
        for u in value.tolist():
            for v in nbrs.get(u, []):
                if v in loc:
                    want.add((loc[u], loc[v]))
        assert int(sub["ptr"][c + 1]) - r0 == num_nodes
        assert sub["node_id"][r0:r0 + len(value)].tolist() == value.tolist()
        assert sub["node_id"][r0 + len(value):r0 + num_nodes].tolist() == [N + k for k in keys]
        assert sub["core"][r0:r0 + num_nodes].tolist() == [True] * len(value) + [False] * len(keys)
    assert got == want


def test_torch_assembly_equals_numpy_assembly():
    """assemble_subgraphs_torch (device tensor ops, chunked) == assemble_subgraphs (NumPy), edge set and row layout."""
    import torch

    from fitgnn_amd.data import assemble_subgraphs, assemble_subgraphs_torch, synthetic_graph

    ei = synthetic_graph(500, 1300, seed=7)
    rng = np.random.default_rng(1)
    assign = rng.integers(0, 120, size=500); assign[:120] = np.arange(120)
    for extra in (True, False):
        a = assemble_subgraphs(ei, 500, assign, 120, extra_node=extra)
        b = assemble_subgraphs_torch(torch.from_numpy(ei), 500, assign, 120, extra_node=extra, chunk_rows=97)
        assert np.array_equal(a["ptr"], b["ptr"].numpy()) and np.array_equal(a["node_id"], b["node_id"].numpy())
        assert np.array_equal(a["core"], b["core"].numpy())
        ea = set(zip(a["edge_index"][0].tolist(), a["edge_index"][1].tolist()))
        eb = set(zip(b["edge_index"][0].tolist(), b["edge_index"][1].tolist()))
        assert ea == eb and len(eb) == b["edge_index"].shape[1]


def test_community_detection_substitute_and_merge():
    """detect_communities finds planted communities; merge_communities == utils.merge_communities (largest first, whole
    communities while the total stays <= k, induced subgraph renumbered in that order)."""
    import torch

    from fitgnn_amd import pipeline

    rng = np.random.default_rng(0)
    sizes, edges, off = [60, 40, 25, 10], [], 0
    for s in sizes:                                        # dense blocks ...
        a = rng.integers(0, s, size=6 * s); b = rng.integers(0, s, size=6 * s)
        k = a != b
        edges.append(np.stack([a[k] + off, b[k] + off])); off += s
    N = off
    ei = np.concatenate(edges + [np.array([[0, 60, 100], [60, 100, 125]])], axis=1)   # ... joined by three single edges
    ei = np.unique(np.concatenate([ei, ei[::-1]], axis=1), axis=1)
    lab = pipeline.detect_communities(ei, N, seed=1)
    truth = np.repeat(np.arange(4), sizes)
    for c in range(4):                                      # every planted block is (almost) one community
        blk = lab[truth == c]
        assert np.bincount(blk).max() >= 0.9 * len(blk)
    data = pipeline.NodeData(torch.arange(N).float().view(-1, 1), ei, torch.from_numpy(truth))
    out = pipeline.merge_communities(data, truth, 75)       # 60 fits, 40 does not, 25 does not (85 > 75), 10 fits: 70 nodes
    assert out.num_nodes == 70 and out.x.flatten().tolist() == list(range(60)) + list(range(125, 135))
    assert int(out.edge_index.max()) < 70 and out.y.tolist() == [0] * 60 + [3] * 10


def test_tile_makers_cover_rows_exactly_once_hypothesis():
    """make_tiles / make_tiles_pair (host side of the SpMM): for random block structures every output row lies in
    exactly one tile, tiles respect the row cap, and a tile's window is the operand range of the blocks it packs."""
    from hypothesis import given, settings, strategies as st

    from fitgnn_amd.csr import make_tiles, make_tiles_pair

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.tuples(st.integers(0, 40), st.integers(0, 40)), min_size=1, max_size=30), st.integers(4, 24))
    def check(blocks, cap):
        out_ptr = np.concatenate([[0], np.cumsum([b[0] for b in blocks])])
        win_ptr = np.concatenate([[0], np.cumsum([b[1] for b in blocks])])
        t = make_tiles_pair(out_ptr, win_ptr, cap)
        cover = np.zeros(int(out_ptr[-1]), dtype=np.int64)
        for tt in t:
            rb, re, wb, wr = int(tt["row_begin"]), int(tt["row_end"]), int(tt["win_begin"]), int(tt["win_rows"])
            assert 0 < re - rb <= cap
            cover[rb:re] += 1
            b0 = int(np.searchsorted(out_ptr, rb, side="right")) - 1          # first block with rows in the tile
            while out_ptr[b0 + 1] == out_ptr[b0] and out_ptr[b0] == rb and b0 + 1 < len(blocks) and win_ptr[b0] < wb:
                b0 += 1
            assert wb in win_ptr and wb + wr in win_ptr                       # windows are whole operand ranges of blocks
        assert np.all(cover == 1)
        sq = make_tiles(out_ptr, cap)                                         # the square case: window == own rows
        cov2 = np.zeros(int(out_ptr[-1]), dtype=np.int64)
        for tt in sq:
            cov2[int(tt["row_begin"]):int(tt["row_end"])] += 1
            assert int(tt["row_end"]) - int(tt["row_begin"]) <= cap
        assert np.all(cov2 == 1)

    check()


def test_batched_dense_prelude_equals_one_at_a_time():
    """coarsening._dense_prelude_batch (components of equal size through one batched eigh) == _dense_prelude per component."""
    import scipy.sparse as sp

    from fitgnn_amd import coarsening as co
    from fitgnn_amd.graph_data import synthetic_molecules

    mol = synthetic_molecules(40, seed=2)
    ei, off = mol["edge_index"], mol["node_ptr"]
    N = int(off[-1])
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
    A = np.zeros((N, 10)); Kc = np.zeros(40, dtype=np.int32)
    co._dense_prelude_batch(W, off, np.arange(40), 10, A, Kc)
    for c in range(40):
        b, e = int(off[c]), int(off[c + 1])
        ref = co._dense_prelude(W, b, e, 10)
        assert Kc[c] == ref.shape[1]
        assert np.array_equal(A[b:e, :ref.shape[1]], ref), c


def test_numpy_assembly_matches_the_reference_loop_on_cora():
    """data.assemble_subgraphs (host NumPy) == the reference's per-cluster loop restated in oracle/gs_oracle.py, on the real
    Cora giant component with the partition the reference recorded (tests/golden): node lists, own / extra flags, edges."""
    from golden_util import Golden

    from fitgnn_amd.data import assemble_subgraphs
    from oracle import gs_oracle

    g = Golden("cora_giant")
    coo = g.W.tocoo()
    ei = np.stack([coo.row, coo.col]).astype(np.int64)
    assign = g.final(0.5)["assign"].astype(np.int64)
    n = int(assign.max()) + 1
    for extra in (True, False):
        sub = assemble_subgraphs(ei, g.N, assign, n, extra_node=extra)
        ref = gs_oracle.cluster_subgraphs(ei, g.N, assign, extra)
        owner = np.searchsorted(sub["ptr"], sub["edge_index"][0], side="right") - 1
        for c in range(0, n, 7):   # every 7th cluster keeps the CPU tier short
            s = ref[c]
            r0, r1 = int(sub["ptr"][c]), int(sub["ptr"][c + 1])
            assert np.array_equal(sub["node_id"][r0:r1], s["orig_idx"])
            assert np.array_equal(sub["core"][r0:r1], ~np.isin(s["orig_idx"], s["actual_ext"]))
            mine = sub["edge_index"][:, owner == c] - r0
            assert set(zip(mine[0].tolist(), mine[1].tolist())) == set(zip(s["edge_index"][0].tolist(), s["edge_index"][1].tolist()))


def test_select_clusters_and_sharding_partition_the_union():
    """data.select_clusters (NumPy and torch forms) keeps whole subgraphs intact; the shards of data.shard_clusters are a
    partition of the union and balance nnz'."""
    from fitgnn_amd import data

    ei = data.synthetic_graph(300, 900, seed=3)
    rng = np.random.default_rng(0)
    _, assign = np.unique(rng.integers(0, 40, size=300), return_inverse=True)
    n = int(assign.max()) + 1
    sub = data.assemble_subgraphs(ei, 300, assign, n, extra_node=True)
    subt = data.assemble_subgraphs_torch(torch.from_numpy(ei), 300, assign, n, extra_node=True)
    nz = data.cluster_nnz(sub)
    assert np.array_equal(nz, data.cluster_nnz(subt))
    assert nz.sum() == sub["edge_index"].shape[1] + sub["ptr"][-1]
    owner = data.shard_clusters(None, nz, 3)
    load = np.bincount(owner, weights=nz, minlength=3)
    assert load.max() - load.min() <= nz.max()
    total_edges = 0
    for r in range(3):
        cl = np.nonzero(owner == r)[0]
        a, b = data.select_clusters(sub, cl), data.select_clusters(subt, cl)
        for k in a:
            assert np.array_equal(np.asarray(a[k]), b[k].numpy()), k
        for j, c in enumerate(cl):
            r0, r1, o0, o1 = a["ptr"][j], a["ptr"][j + 1], sub["ptr"][c], sub["ptr"][c + 1]
            assert np.array_equal(a["node_id"][r0:r1], sub["node_id"][o0:o1])
            ea = a["edge_index"][:, (a["edge_index"][0] >= r0) & (a["edge_index"][0] < r1)] - r0
            eo = sub["edge_index"][:, (sub["edge_index"][0] >= o0) & (sub["edge_index"][0] < o1)] - o0
            assert np.array_equal(ea, eo)
        total_edges += a["edge_index"].shape[1]
    assert total_edges == sub["edge_index"].shape[1]


@pytest.mark.parametrize("layout", ["sorted", "star"])
def test_a_rank_assembles_only_its_own_clusters(layout):
    """Data parallel set-up (bench.py): the partition is sharded BEFORE any subgraph exists, on a weight every rank
    computes from the graph and the partition (data.cluster_weights_torch: a lower bound of nnz' that leaves out the edges
    between extra nodes), and assemble_subgraphs_torch(clusters=mine) builds the rank's subgraphs alone -- the same dict as
    cutting them out of the whole union (data.select_clusters), without ever building the others."""
    from fitgnn_amd import data

    ei = data.synthetic_graph(3000, 9000, seed=3)
    rng = np.random.default_rng(0)
    assign = rng.integers(0, 400, size=3000)
    assign[:400] = np.arange(400)
    e, a = torch.from_numpy(ei), torch.from_numpy(assign)
    full = data.assemble_subgraphs_torch(e, 3000, a, 400, layout=layout)
    nnz = data.cluster_nnz(full)
    w = data.cluster_weights_torch(e, 3000, a, 400)
    assert np.all(w <= nnz) and np.all(w >= 0.5 * nnz), "a lower bound that tracks nnz'"
    owner = data.shard_clusters(None, w, 3)
    load = np.bincount(owner, weights=nnz, minlength=3)
    assert load.max() <= 1.1 * load.min(), "shards balanced in the real nnz' too"
    rows = 0
    for r in range(3):
        mine = np.nonzero(owner == r)[0]
        want = data.select_clusters(full, mine)
        got = data.assemble_subgraphs_torch(e, 3000, a, 400, layout=layout, clusters=mine)
        assert set(want) == set(got)
        for k in want:
            if k == "edge_index":   # the same edge set (the enumeration order follows the member rows either way)
                assert torch.equal(want[k], got[k])
            else:
                assert torch.equal(want[k], got[k]), k
        rows += int(got["ptr"][-1])
    assert rows == int(full["ptr"][-1])


def test_star_layout_is_a_row_permutation_of_the_sorted_layout():
    """assemble_subgraphs_torch(layout="star") holds the same subgraphs as the reference's sorted layout -- same member set per
    cluster, same edges between the same (cluster, node) pairs -- with every subgraph's rows ordered star by star: an own node,
    then the extra nodes whose lowest own neighbour it is; seg_start marks the first row of every star."""
    from fitgnn_amd.data import assemble_subgraphs_torch, select_clusters, synthetic_graph

    N = 700
    ei = synthetic_graph(N, 2400, seed=11)
    rng = np.random.default_rng(2)
    _, assign = np.unique(rng.integers(0, 150, size=N), return_inverse=True)
    n = int(assign.max()) + 1
    a = assemble_subgraphs_torch(torch.from_numpy(ei), N, assign, n, extra_node=True)
    b = assemble_subgraphs_torch(torch.from_numpy(ei), N, assign, n, extra_node=True, layout="star", chunk_rows=333)
    assert torch.equal(a["ptr"], b["ptr"]) and "seg_start" in b and "seg_start" not in a
    adj = {}
    for u, v in zip(ei[0].tolist(), ei[1].tolist()):
        adj.setdefault(u, set()).add(v)
    ptr = a["ptr"].numpy()
    pair = lambda d: set(zip(d["node_id"][d["edge_index"][0]].tolist(), d["node_id"][d["edge_index"][1]].tolist(),
                            (torch.searchsorted(d["ptr"], d["edge_index"][0], right=True) - 1).tolist()))
    assert pair(a) == pair(b) and a["edge_index"].shape == b["edge_index"].shape
    for c in range(n):
        r0, r1 = int(ptr[c]), int(ptr[c + 1])
        ids, core, seg = b["node_id"][r0:r1].tolist(), b["core"][r0:r1].tolist(), b["seg_start"][r0:r1].tolist()
        assert sorted(ids) == a["node_id"][r0:r1].tolist()
        assert [i for i, k in zip(ids, core) if k] == sorted(np.nonzero(assign == c)[0].tolist()), "own nodes ascending, each opening a star"
        own = set(np.nonzero(assign == c)[0].tolist())
        hub = None
        for i, k, s0 in zip(ids, core, seg):
            if k:
                assert s0
                hub, last = i, -1
            else:
                assert not s0 and i > last and hub == min(adj[i] & own), "an extra node sits in the star of its lowest own neighbour"
                last = i
    # a shard keeps the star marks of its rows
    sh = select_clusters(b, np.arange(0, n, 3))
    assert int(sh["seg_start"].sum()) == int(sh["core"].sum())


def test_stream_ranges_cover_whole_segments():
    """csr.stream_ranges: the segment-streaming SpMM kernel's work list -- every range is a run of WHOLE segments, the ranges
    partition the segments in order, carry about equal row counts, and empty segments vanish."""
    from fitgnn_amd.csr import stream_ranges

    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.integers(1, 120, size=5000), [0, 0, 900, 1, 0]])
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    n = int(ptr[-1])
    seg, rs = (t.numpy() for t in stream_ranges(ptr, n, "cpu", want=64, min_rows=64))
    assert seg[0] == 0 and seg[-1] == n and np.all(np.diff(seg) > 0)
    assert set(seg.tolist()) == set(ptr.tolist())
    assert rs[0] == 0 and rs[-1] == len(seg) - 1 and np.all(np.diff(rs) >= 0) and len(rs) == 65
    rows = np.diff(seg[rs])
    assert rows.sum() == n and rows.max() <= n / 64 + 900
    # fewer rows than ranges asked for: one range per 64 rows at most, never more ranges than segments
    seg, rs = stream_ranges(np.array([0, 3, 10]), 10, "cpu")
    assert rs.tolist() == [0, 2]
    with pytest.raises(ValueError):
        stream_ranges(np.array([0, 3]), 10, "cpu")


def test_two_hop_index_covers_every_entry_the_kernel_cannot_serve_from_lds():
    """ops._two_hop_block_index (the side-table index of fitgnn_spmm_two_hop_blocks_f32) on the CPU, against a literal restatement of
    which entries the whole-subgraph kernel serves from LDS -- row and column in the same 16-row piece of the same block, the column a
    carried long row of the row's block (the first four of the block's long rows), or the row itself one: every other entry's
    column has a table row, as have the loss rows, the carried long rows, the rows outside the blocks and the rows with two or more
    loss columns; every remaining ("simple") row has at most one loss column and row_p / row_w name it."""
    import types
    import torch
    from fitgnn_amd import csr, ops

    rng = np.random.default_rng(5)
    sizes = [40, 7, 120, 16, 17, 300, 3, 64]
    src, dst, off = [], [], 0
    for s_ in sizes:   # stars with three centres (rows 0..2 of a block) and some leaf -- leaf edges
        for h in range(min(3, s_ - 1)):
            leaves = np.arange(3, s_) if s_ > 3 else np.arange(1, s_)
            src += [off + h] * len(leaves) + (off + leaves).tolist()
            dst += (off + leaves).tolist() + [off + h] * len(leaves)
        if s_ > 8:
            a, b = rng.integers(0, s_, size=s_), rng.integers(0, s_, size=s_)
            k = a != b
            src += (off + a[k]).tolist() + (off + b[k]).tolist()
            dst += (off + b[k]).tolist() + (off + a[k]).tolist()
        off += s_
    n = off
    e = np.unique(np.array([src + list(range(n)), dst + list(range(n))]), axis=1)   # with self loops, (row, col) sorted
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, e[0] + 1, 1)
    rowptr = np.cumsum(rowptr)
    col = e[1]
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    small, blocks, long_rows = csr.split_blocks(ptr, torch.from_numpy(rowptr.astype(np.int32)), 16, 4096)
    assert len(blocks) and len(long_rows)
    side = types.SimpleNamespace(rowptr=torch.from_numpy(rowptr.astype(np.int32)), col=torch.from_numpy(col.astype(np.int32)),
                                 val=torch.from_numpy(rng.random(len(col)).astype(np.float32)), blocks=torch.from_numpy(blocks),
                                 long_rows=torch.from_numpy(long_rows if len(long_rows) else np.zeros(1, np.int32)))
    g = types.SimpleNamespace(t=side, n=n)
    rows = torch.from_numpy(np.sort(rng.choice(n, size=n // 5, replace=False)).astype(np.int64))
    n_sel = int(rows.numel())
    pos = n_sel + torch.arange(n, dtype=torch.int32) % ops.ZERO_ROWS
    pos[rows] = torch.arange(n_sel, dtype=torch.int32)
    ix = ops._two_hop_block_index(g, rows, pos)
    zrow, zcol, zt_rows = ix["zrow"].numpy(), ix["zcol"].numpy(), ix["zt_rows"].numpy()
    assert np.array_equal(zt_rows[:n_sel], rows.numpy()) and len(set(zt_rows.tolist())) == len(zt_rows)
    assert np.array_equal(zrow[zt_rows], np.arange(len(zt_rows))) and int((zrow >= 0).sum()) == len(zt_rows)
    # the kernel's rule, literally
    block_of, piece_of, carried = np.full(n, -1), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=bool)
    for b, rec in enumerate(blocks):
        r0, r1, _, _, loff, nl = rec[:6]
        block_of[r0:r1] = b
        piece_of[r0:r1] = (np.arange(r0, r1) - r0) // 16
        carried[long_rows[loff:loff + min(nl, 4)]] = True
    is_loss = np.zeros(n, dtype=bool); is_loss[rows.numpy()] = True
    for r in range(n):
        cols = col[rowptr[r]:rowptr[r + 1]]
        loss_cols = [c for c in cols if is_loss[c]]
        for j, c in enumerate(cols):
            served = block_of[r] >= 0 and block_of[r] == block_of[c] and (piece_of[r] == piece_of[c] or carried[c] or carried[r])
            zc = zcol[rowptr[r] + j]
            assert zc == (zrow[c] if zrow[c] >= 0 else ops.NO_ROW)
            assert served or zc != ops.NO_ROW, (r, c)
            assert not is_loss[c] or zc == pos[c]   # a loss column's table row is its compact position (its operand row)
        if block_of[r] < 0 or carried[r] or is_loss[r] or len(loss_cols) >= 2:
            assert zrow[r] >= 0, r
        if zrow[r] < 0:   # a simple row
            assert len(loss_cols) <= 1
            if loss_cols:
                j = list(cols).index(loss_cols[0])
                assert ix["row_p"][r] == pos[loss_cols[0]] and float(ix["row_w"][r]) == float(side.val[rowptr[r] + j])
            else:
                assert int(ix["row_p"][r]) == ops.NO_ROW and float(ix["row_w"][r]) == 0.0
    assert np.array_equal(np.sort(ix["tile_zt"].numpy()), np.sort(zrow[block_of < 0]))


def test_host_tile_packing_and_block_split_equal_their_python_reference():
    """fitgnn_make_tiles_host / fitgnn_split_blocks_host (the library's host code: 165 000 stars -> 515 000 tiles in milliseconds)
    against the Python loops they replace, on random block structures: the same tiles, block records and long-row lists."""
    from fitgnn_amd import csr

    rng = np.random.default_rng(11)
    for trial in range(40):
        nb = int(rng.integers(0, 60))
        sizes = rng.choice([1, 2, 3, 5, 9, 15, 16, 17, 30, 64, 200, 1000], size=nb)
        ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        cap = int(rng.choice([4, 16, 64]))
        a, b = csr.make_tiles(ptr, cap), csr.make_tiles_py(ptr, cap)
        assert a.dtype == b.dtype and np.array_equal(a, b), (trial, "tiles")
        n = int(ptr[-1])
        deg = rng.integers(1, 40, size=n)
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        limit = int(rng.choice([4 * cap, 100, 5000]))
        s1, b1, l1 = csr.split_blocks(ptr, torch.from_numpy(rowptr), cap, limit)
        s2, b2, l2 = csr.split_blocks_py(ptr, torch.from_numpy(rowptr), cap, limit)
        assert np.array_equal(s1, s2), (trial, "small tiles")
        assert np.array_equal(b1, b2), (trial, "blocks")
        assert np.array_equal(l1, l2), (trial, "long rows")


def test_tiles_cut_at_graph_boundaries_partition_every_graph():
    """graph_data.cut_tiles_at_graphs (the device batch assembly's tile table): the packed tiles of a dataset's diagonal blocks cut at every
    graph boundary -- a partition of the rows in order, no tile across two graphs, every tile a run of whole blocks of at most `cap`
    rows (or a piece of a larger block) whose window is its own rows, per-graph tile ranges that cover exactly the graph's rows."""
    from fitgnn_amd import csr, graph_data

    rng = np.random.default_rng(5)
    for trial in range(30):
        n_graphs = int(rng.integers(1, 25))
        blocks_per_graph = rng.integers(1, 7, size=n_graphs)
        sizes = rng.choice([1, 2, 3, 5, 9, 15, 16, 17, 40], size=int(blocks_per_graph.sum()))
        ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        g_row = ptr[np.concatenate([[0], np.cumsum(blocks_per_graph)])]
        R = int(ptr[-1])
        rowptr = np.concatenate([[0], np.cumsum(rng.integers(1, 9, size=R))]).astype(np.int64)
        cap = int(rng.choice([4, 16]))
        tiles, g_tile = graph_data.cut_tiles_at_graphs(csr.make_tiles(ptr, cap), g_row, rowptr)
        assert tiles[0, 0] == 0 and tiles[-1, 1] == R and np.array_equal(tiles[1:, 0], tiles[:-1, 1])
        assert np.array_equal(tiles[:, 2], tiles[:, 0]) and np.array_equal(tiles[:, 3], tiles[:, 1] - tiles[:, 0])
        assert np.all(tiles[:, 3] <= cap) and np.all(tiles[:, 3] > 0)
        assert np.array_equal(tiles[:, 4], rowptr[tiles[:, 0]]) and np.array_equal(tiles[:, 5], rowptr[tiles[:, 1]])
        block_set = set(ptr.tolist())
        for a, b in tiles[:, :2]:
            whole = (int(a) in block_set) and (int(b) in block_set)
            inside = np.searchsorted(ptr, a, side="right") == np.searchsorted(ptr, b - 1, side="right")
            assert whole or inside, (trial, a, b)   # a run of whole blocks, or a piece inside one large block
        assert g_tile[0] == 0 and g_tile[-1] == len(tiles)
        for g in range(n_graphs):
            t = tiles[g_tile[g]:g_tile[g + 1]]
            assert len(t) > 0 and t[0, 0] == g_row[g] and t[-1, 1] == g_row[g + 1]


def test_appnp_plan_sends_every_row_to_exactly_one_launch():
    """ops.AppnpPlan (host logic over the library's size queries; no GPU): the diagonal blocks of a block-diagonal pattern are split
    into units (packed to <= 64 rows), LDS launches (one per slice width and workgroup size, each within the kernel's item and LDS
    limits for ITS largest range), the global-scratch tier (only with enough blocks) and the open rows -- every row exactly once."""
    import numpy as np
    import torch
    from fitgnn_amd import _lib, csr, ops

    L = _lib.lib()
    sizes = [5] * 40 + [64, 65, 100, 300, 70, 1500, 2500, 1024, 1025, 2048, 4096, 5000, 3, 7] + [900] * 10
    src, dst, off = [], [], 0
    for sz in sizes:   # rings
        r = np.arange(sz)
        a, b = r, np.roll(r, -1)
        m = a != b
        src += [a[m] + off, b[m] + off]; dst += [b[m] + off, a[m] + off]
        off += sz
    ei = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)]))
    g = csr.CSRGraph(ei, off, mode="gcn")
    for h4 in (1, 12, 16):
        plan = ops.AppnpPlan(g, h4)
        cover = np.zeros(off, dtype=np.int64)
        for a, b in plan.units.numpy():
            cover[a:b] += 1
        assert plan.max_rows <= plan.cap_rows and plan.max_rows * plan.unit_slice <= 4 * plan.unit_threads
        assert L.fitgnn_appnp_lds_bytes(plan.max_rows, plan.max_entries, plan.unit_slice) <= L.fitgnn_appnp_lds_max_bytes()
        seen = set()
        for sl, ranges, m, mr, me, threads in plan.lds_launches:
            assert (sl, threads) not in seen and sl <= h4 and m == len(ranges)
            seen.add((sl, threads))
            assert mr * sl <= 4 * threads and L.fitgnn_appnp_lds_bytes(mr, me, sl) <= L.fitgnn_appnp_lds_max_bytes()
            rr = ranges.numpy()
            assert (rr[:, 1] - rr[:, 0]).max() == mr
            for a, b in rr:
                cover[a:b] += 1
        assert plan.n_blocks == 0 and plan.blocks is None   # fewer than MIN_BLOCKS blocks beyond LDS
        if plan.n_open:
            cover[plan.open_rows.numpy()] += 1
        assert (cover == 1).all()
        assert plan.rows_in_units + plan.rows_in_lds_blocks + plan.n_open == off
        # open: the block beyond four items per thread x 1 024 threads at a one-float4 slice, and the 4 096-row one, whose CSR slice
        # (three entries a row) and buffer exceed LDS even at that slice
        assert plan.n_open == 5000 + 4096
        old = ops.AppnpPlan(g, h4, sliced=False)
        assert old.lds_launches == [] and old.rows_in_units == plan.rows_in_units and old.n_open == off - old.rows_in_units
