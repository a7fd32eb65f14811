"""CPU oracle for the callers either side of the hot path (TEST INFRASTRUCTURE ONLY; never imported by fit-gnn_amd/):
the per-cluster subgraph construction and the coarsened-graph (Gc) label / mask / edge assembly, restated literally
from the reference's Python -- per-cluster loops, NumPy set operations, scipy `C.dot` -- so that the vectorised /
device implementations in fitgnn_amd.data and fitgnn_amd.pipeline can be compared with it on small inputs.

Follows, as text:
    utils.py:52-62      neighbour / nodes_2_neighbours (edge scans)
    utils.py:186-267    per-cluster subgraph: value = cluster nodes (+ 1-hop neighbours with --extra_node), sorted;
                        M = data.subgraph(value) (PyG: induced subgraph, nodes relabelled by position in `value`,
                        edges kept in the order of data.edge_index); M.mask; M.map_dict
    utils.py:683-703    per-subgraph train / val / test masks (False on extra nodes)
    utils.py:705-775    Gc: pooled features C.X, pooled one-hot labels -> argmax label, single-class mask rule
                        (:726-730), Gc.W edges with running node offset (:744-752), pass-through of components with
                        <= 10 nodes that hold train / val nodes (:754-769), 'does not need coarsening' (:763)
PARITY: the arithmetic here is integer / set logic plus scipy sparse products, whose order the reference fixes; the
inputs the tests feed it (C, Gc.W) are the reference's own recorded outputs (tests/golden/coarsen_*.npz).
`torch_geometric.data.Data.subgraph` is third party and absent: restated from its documented behaviour.
"""
import numpy as np
import scipy.sparse as sp


def neighbour(edge_index, node):
    """utils.py:52-56."""
    hit = np.nonzero(edge_index[0] == node)[0]
    return edge_index[1][hit].flatten()


def nodes_2_neighbours(edge_index, nodes):
    """utils.py:58-62."""
    mask = np.isin(edge_index[0], nodes)
    return np.unique(edge_index[1, mask])


def data_subgraph(edge_index, subset):
    """PyG Data.subgraph(subset) on the edge list: keep edges with both ends in `subset`, in input order, relabel
    each end by its position in `subset` (subset is sorted and duplicate free here, utils.py:243)."""
    subset = np.asarray(subset, dtype=np.int64)
    pos = {int(v): i for i, v in enumerate(subset)}
    keep = np.isin(edge_index[0], subset) & np.isin(edge_index[1], subset)
    e = edge_index[:, keep]
    return np.array([[pos[int(a)] for a in e[0]], [pos[int(b)] for b in e[1]]], dtype=np.int64).reshape(2, -1)


def cluster_subgraphs(edge_index, num_nodes, assign, extra_node):
    """utils.py:186-267 for node_cls without --cluster_node: one subgraph per cluster, in ascending cluster id (the
    reference walks meta_node_2_node in order of first appearance over ascending component nodes, and a cluster's id is
    the rank of its smallest member, coarsening_utils.py:168-179, so the two orders agree).
    Returns a list of dicts: orig_idx (sorted node ids), actual_ext (extra nodes), edge_index (relabelled), mask
    (the reference's M.mask: first len(cluster) positions True -- quirk (iii) of SURVEY §8 a12), map_dict."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    assign = np.asarray(assign, dtype=np.int64)
    out = []
    for c in range(int(assign.max()) + 1 if len(assign) else 0):
        value = np.sort(np.nonzero(assign == c)[0])
        actual_ext = np.array([], dtype=np.int64)
        if extra_node:
            extra = nodes_2_neighbours(edge_index, value)                  # :235
            actual_ext = extra[~np.isin(extra, value)]                     # :237
            value = np.concatenate([value, actual_ext])                    # :238
        value = np.sort(value)                                             # :243
        mapping = {int(v): i for i, v in enumerate(value)}                 # :245-247
        n_own = len(value) - len(actual_ext)
        mask = np.array([True] * n_own + [False] * len(actual_ext), dtype=bool)   # :261 (positions, not membership)
        out.append(dict(orig_idx=value, actual_ext=actual_ext, edge_index=data_subgraph(edge_index, value), mask=mask,
                        map_dict=mapping))
    return out


def subgraph_split_masks(sub, train_mask, val_mask, test_mask, extra_node):
    """utils.py:683-703: the dataset masks stamped onto a subgraph through map_dict, cleared on extra nodes."""
    n = len(sub["orig_idx"])
    tr, va, te = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
    ext = set(int(v) for v in sub["actual_ext"])
    for node, new_node in sub["map_dict"].items():
        if train_mask[node]:
            tr[new_node] = True
        if val_mask[node]:
            va[new_node] = True
        if test_mask[node]:
            te[new_node] = True
        if extra_node and node in ext:
            tr[new_node] = va[new_node] = te[new_node] = False
    return tr, va, te


def one_hot(x, class_count):
    """utils.py:48-49."""
    return np.eye(class_count, dtype=np.float32)[np.asarray(x, dtype=np.int64), :]


def load_gc(components, C_list, GcW_list, features, labels, train_mask, val_mask, n_classes):
    """utils.py:705-775.  components: list of (orig_idx list, W csr) sorted by size descending (utils.py:146);
    C_list / GcW_list: scipy C [n x N_H] and Gc.W of the components with more than 10 nodes, in that order (:164-166).
    Returns dict(features f32 [n, F], train_labels int64, train_mask bool, val_labels, val_mask, edge int64 [2, E])."""
    feats, tl, tm, vl, vm = [], [], [], [], []
    rows = cols = None
    coarsen_node = 0
    for number, (orig_idx, HW) in enumerate(components):
        keep = np.asarray(orig_idx, dtype=np.int64)
        H_features, H_labels = features[keep], labels[keep]
        H_train, H_val = train_mask[keep], val_mask[keep]
        if len(keep) > 10 and H_train.sum() + H_val.sum() > 0:
            train_labels = one_hot(H_labels, n_classes)
            train_labels[~H_train] = 0
            val_labels = one_hot(H_labels, n_classes)
            val_labels[~H_val] = 0
            C, GcW = sp.csc_matrix(C_list[number]), sp.csr_matrix(GcW_list[number])   # :723-724 (indexed by rank)
            new_masks = []
            for lab in (train_labels, val_labels):
                pooled = C.dot(lab)                                       # f64 [n, classes]
                new_mask = np.sum(pooled, axis=1).astype(bool)            # torch.BoolTensor(np.sum(...)) :726
                mix = pooled.astype(np.float32)
                mix[mix > 0] = 1
                new_mask[np.sum(mix, axis=1) > 1] = False                 # :727-730
                new_masks.append(new_mask)
            feats.append(C.dot(H_features).astype(np.float32))            # torch.FloatTensor(C.dot(H_features)) :738
            tl.append(np.argmax(C.dot(train_labels).astype(np.float32), axis=1))
            tm.append(new_masks[0])
            vl.append(np.argmax(C.dot(val_labels).astype(np.float32), axis=1))
            vm.append(new_masks[1])
            coo = GcW.tocoo()
            r, c = coo.row.astype(np.int64) + coarsen_node, coo.col.astype(np.int64) + coarsen_node
            rows, cols = (r, c) if rows is None else (np.concatenate([rows, r]), np.concatenate([cols, c]))
            coarsen_node += GcW.shape[0]
        elif H_train.sum() + H_val.sum() > 0:
            feats.append(np.asarray(H_features, dtype=np.float32))
            tl.append(H_labels); tm.append(H_train); vl.append(H_labels); vm.append(H_val)
            if rows is None:
                raise Exception("The graph does not need coarsening.")    # :763
            coo = sp.csr_matrix(HW).tocoo()
            rows = np.concatenate([rows, coo.row.astype(np.int64) + coarsen_node])
            cols = np.concatenate([cols, coo.col.astype(np.int64) + coarsen_node])
            coarsen_node += HW.shape[0]
    return dict(features=np.concatenate(feats), train_labels=np.concatenate(tl).astype(np.int64), train_mask=np.concatenate(tm),
                val_labels=np.concatenate(vl).astype(np.int64), val_mask=np.concatenate(vm), edge=np.stack([rows, cols]))
