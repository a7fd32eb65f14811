"""Flat, memory-mappable on-disk form of the coarsening artefacts.

The reference pickles pygsp Graph objects, scipy matrices and lists of PyG Data per cluster
(main.py:131-172 `save`, loaded back at main.py:270-275 / inference.py:543-548).  Here an artefact is a directory of raw
little-endian `.npy` arrays plus `meta.json`: int32 CSR / index arrays, f32 features, bool masks, the int32 node ->
cluster assignment and the f64 non-zeros of C.  `np.load(..., mmap_mode="r")` maps them without reading, and every array
goes to the device with one copy, so a 165 000-node community (8.2 M union rows) is a handful of large sequential reads
instead of 82 500 unpickled objects.  File naming follows the reference: <ratio>_<node_type>_<graph_type>/.
"""
import json
import os

import numpy as np
import torch

FORMAT = "fitgnn-flat-1"


def artefact_dir(root, args):
    """main.py:134-143: node_type d / e / c (default / --extra_node / --cluster_node), graph_type full / community."""
    node_type = "e" if getattr(args, "extra_node", False) else ("c" if getattr(args, "cluster_node", False) else "d")
    graph_type = "community" if getattr(args, "use_community_detection", False) else "full"
    return os.path.join(root, f"{args.coarsening_ratio}_{node_type}_{graph_type}")


def _np(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a)


def save_arrays(path, arrays, meta=None):
    os.makedirs(path, exist_ok=True)
    index = {}
    for name, a in arrays.items():
        if a is None:
            continue
        a = _np(a)
        np.save(os.path.join(path, name + ".npy"), a)
        index[name] = {"dtype": str(a.dtype), "shape": list(a.shape)}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"format": FORMAT, "arrays": index, **(meta or {})}, f, indent=1)


def load_arrays(path, mmap=True):
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT:
        raise ValueError(f"{path}: not a {FORMAT} artefact")
    arrays = {}
    for name, info in meta["arrays"].items():
        a = np.load(os.path.join(path, name + ".npy"), mmap_mode="r" if mmap else None)
        if list(a.shape) != info["shape"] or str(a.dtype) != info["dtype"]:
            raise ValueError(f"{path}/{name}.npy does not match meta.json")
        arrays[name] = a
    return arrays, meta


def save_gs(path, batch, coarsened=None, extra_meta=None):
    """A data.SubgraphBatch (+ the node -> cluster assignment it was built from) as a flat artefact.  Features are
    stored once per ORIGINAL node (`x_table`) when the batch carries the de-duplicated table, else per union row."""
    arrays = dict(ptr=np.asarray(batch.ptr, dtype=np.int64), node_id=batch.node_id.to(torch.int32),
                  core=batch.core, edge_index=batch.edge_index.to(torch.int32), y=batch.y,
                  train_mask=batch.train_mask)
    if batch.x_table is None:
        raise ValueError("save_gs needs the batch's feature table (SubgraphBatch(dedup=True), the default on the device)")
    arrays["x_table"] = batch.x_table
    for k in ("val_idx", "test_idx"):
        if hasattr(batch, k):
            arrays[k] = getattr(batch, k)
    if coarsened is not None:
        arrays["assign"] = np.asarray(coarsened.assign, dtype=np.int32)
    save_arrays(path, arrays, dict(kind="gs", n_rows=int(batch.n_rows), nnz_prime=int(batch.nnz), **(extra_meta or {})))


def load_gs(path, device="cuda"):
    """Inverse of save_gs: a device-resident SubgraphBatch (CSR and tiles are rebuilt on the device; they are derived
    data).  The arrays are memory-mapped and copied to the device once."""
    from . import data as fdata

    arrays, meta = load_arrays(path)
    if meta.get("kind") != "gs":
        raise ValueError(f"{path}: kind {meta.get('kind')!r}, expected 'gs'")
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.array(a))  # noqa: E731  (materialise the mapped pages once)
    node_id = t(arrays["node_id"]).long()
    sub = dict(ptr=np.array(arrays["ptr"]), node_id=node_id, core=t(arrays["core"]), edge_index=t(arrays["edge_index"]).long())
    X = t(arrays["x_table"])
    y_tab = torch.zeros(X.shape[0], dtype=torch.long)
    y_tab[node_id] = t(arrays["y"])
    tm_tab = torch.zeros(X.shape[0], dtype=torch.bool)
    tm_tab[node_id[t(arrays["train_mask"])]] = True
    batch = fdata.SubgraphBatch(sub, X, y_tab, tm_tab, device=dev, dedup=True)
    batch.train_mask = t(arrays["train_mask"]).to(dev)              # the exact per-row mask
    batch.train_idx = torch.nonzero(batch.train_mask).flatten()
    batch.y = t(arrays["y"]).to(dev)
    for k in ("val_idx", "test_idx"):
        if k in arrays:
            setattr(batch, k, t(arrays[k]).to(dev))
    batch.assign = np.array(arrays["assign"]) if "assign" in arrays else None
    return batch, meta
