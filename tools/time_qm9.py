#!/usr/bin/env python3
"""GPU-box tool: build + train timing of the graph-level path on the QM9-shaped stand-in (args: n_graphs)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import os as _os
from fitgnn_amd import ops as _ops
if _os.environ.get("FITGNN_NO_HAND_GEMM"):   # A/B: the library GEMMs instead of csrc/gemm_*.hip
    _ops.ATB_KERNEL = _ops.NT_KERNEL = False
from fitgnn_amd import graph_data, network, train

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.time(); mol = graph_data.synthetic_molecules(n, seed=0); t1 = time.time()
print(f"generate {n} molecules: {t1 - t0:.1f} s, nodes {mol['node_ptr'][-1]}, directed edges {mol['edge_index'].shape[1]}", flush=True)
gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda"); torch.cuda.synchronize(); t2 = time.time()
print(f"GraphSet (coarsen_batch + pooling + assembly): {t2 - t1:.1f} s; clusters {gset.co.n_clusters}, levels hist {np.bincount(gset.co.levels).tolist()}, "
      f"Gs union rows {int(gset.sub_ptr[-1])}", flush=True)
args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=512, num_classes=1)
model = network.Regress_graph_gs(args).cuda()
tr = train.GraphTrainer(model, gset, list(range(n // 2)), kind="gs", batch_size=128, lr=0.001); torch.cuda.synchronize(); t3 = time.time()
print(f"trainer ({len(tr.batches)} batches of 128 graphs, CSR per batch): {t3 - t2:.1f} s", flush=True)
tr.step(); torch.cuda.synchronize()
t4 = time.time(); l = float(tr.step()); torch.cuda.synchronize(); t5 = time.time()
model2 = network.Regress_graph_gs(args).cuda()
tc = time.time(); tr2 = train.GraphTrainer(model2, gset, list(range(n // 2)), kind="gs", batch_size=128, lr=0.001, capture=True)
tr2.step(); torch.cuda.synchronize(); tc1 = time.time()
tr2.step(); torch.cuda.synchronize()
tc2 = time.time(); l2 = float(tr2.step()); torch.cuda.synchronize(); tc3 = time.time()
print(f"hipGraph-captured steps: build + capture {tc1 - tc:.1f} s; epoch {tc3 - tc2:.3f} s -> {(n // 2) / (tc3 - tc2):.0f} graphs/s, loss {l2:.4f}", flush=True)
nnzp = int(gset.gs_edge_index.shape[1] * 0.5) + int(gset.sub_ptr[gset.cluster_ptr[n // 2]])
print(f"epoch (Gs, {n // 2} graphs): {t5 - t4:.2f} s -> {(n // 2) / (t5 - t4):.0f} graphs/s, loss {l:.4f}", flush=True)
