#!/usr/bin/env python3
"""Generate tests/golden/gcn_cora_sggc.npz -- the train half's only reference-held known answer.

The reference ships a trained 2-layer PyG GCN for Cora (Baselines/SGGC/GCN/params/checkpoint-best-acc.pkl, written by
Baselines/SGGC/GCN/train.py:93; model = Baselines/SGGC/GCN/network.py: conv1 -> relu -> dropout -> conv2 ->
log_softmax, hidden 64) and the Planetoid raw files of Cora (Baselines/SGGC/APPNP/dataset/cora/raw).  Those are DATA
files; this script copies their contents into one small fixture:
    reference-derived : the four checkpoint tensors (names, shapes, values), Cora x (sparse), edge_index, y, test index
    restatement-derived: `logits` / `test_acc` = oracle/gnn_oracle.py's GCNConv applied to them in eval mode
                         (train.py:101-104: features L1-normalised, full graph, accuracy on the Planetoid test split)
What this pins: the parameter layout PyG's GCNConv saves (`<conv>.lin.weight [out,in]`, `<conv>.bias [out]`) and that
the restatement, fed real trained weights and the real graph, classifies far above chance (0.732 on the Planetoid test
split; chance = 1/7).  What it does NOT pin: GCNConv's numerics -- the checkpoint's training run (ratio, split) is
not recorded and variants of the normalisation score within +-0.03 of each other (measured when this fixture was
made: gcn_norm 0.732, without self loops 0.714, D^-1 A 0.754, plain sum 0.705, no bias 0.702), so the train half stays
"parity unpinned" against PyG; the fixture's `logits` are a frozen output of the restatement on real data.
Run here only (needs /root/reference):  python tests/golden/make_gcn_golden.py
"""
import os
import pickle
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/Baselines/SGGC"


def read_cora(root):
    def rd(s):
        with open(os.path.join(root, f"ind.cora.{s}"), "rb") as f:
            return pickle.load(f, encoding="latin1")
    x, tx, allx, y, ty, ally, graph = (rd(s) for s in ("x", "tx", "allx", "y", "ty", "ally", "graph"))
    test_idx = np.loadtxt(os.path.join(root, "ind.cora.test.index"), dtype=np.int64)
    order = np.sort(test_idx)
    feats = sp.vstack([allx, tx]).tolil()
    feats[test_idx, :] = feats[order, :]
    labels = np.vstack([ally, ty])
    labels[test_idx, :] = labels[order, :]
    pairs = {(u, v) for u, nb in graph.items() for v in nb if u != v}
    pairs |= {(v, u) for (u, v) in pairs}
    ei = np.array(sorted(pairs), dtype=np.int32).T
    return sp.csr_matrix(feats, dtype=np.float32), ei, labels.argmax(1).astype(np.int32), test_idx.astype(np.int32)


def main():
    from oracle import gnn_oracle as G
    sd = torch.load(f"{REF}/GCN/params/checkpoint-best-acc.pkl", map_location="cpu", weights_only=False)
    X, ei, y, test_idx = read_cora(f"{REF}/APPNP/dataset/cora/raw")
    coo = X.tocoo()
    x = torch.from_numpy(X.toarray())
    x = torch.nn.functional.normalize(x, p=1)
    e = torch.from_numpy(ei.astype(np.int64))
    h = torch.relu(G.gcn_conv(x, e, sd["conv1.lin.weight"], sd["conv1.bias"]))
    logits = torch.log_softmax(G.gcn_conv(h, e, sd["conv2.lin.weight"], sd["conv2.bias"]), dim=1)
    pred = logits.argmax(1).numpy()
    acc = float((pred[test_idx] == y[test_idx]).mean())
    print("nodes", X.shape, "edges", ei.shape, "test acc", acc)
    out = {"x_row": coo.row.astype(np.int32), "x_col": coo.col.astype(np.int32), "x_val": coo.data.astype(np.float32),
           "x_shape": np.array(X.shape, dtype=np.int64), "edge_index": ei, "y": y, "test_idx": test_idx,
           "logits": logits.numpy().astype(np.float32), "test_acc": np.float64(acc),
           "keys": np.array(list(sd.keys()))}
    for k, v in sd.items():
        out["w:" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "gcn_cora_sggc.npz"), **out)


if __name__ == "__main__":
    main()
