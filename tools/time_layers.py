#!/usr/bin/env python3
"""GPU-box tool: forward+backward step time of the other layer families on the S-pubmed union (2 layers, hidden 512):
GCNConv (the benchmarked path), GATConv, SAGEConv, GINConv, and APPNP (MLP + K=10 propagation, alpha=0.1)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import bench
from fitgnn_amd import network, nn as fnn

dev = torch.device("cuda")
batch, (Fdim, C), info = bench.build_workload("S-pubmed", 0, dev)
x, ei, y, idx = batch.x, batch.edge_index, batch.y, batch.train_idx
def timeit(step, n=10):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for layer in ("GCNConv", "GATConv", "SAGEConv", "GINConv"):
    args = argparse.Namespace(num_layers1=2, layer_name=layer, num_features=Fdim, hidden=512, num_classes=C)
    torch.manual_seed(0)
    m = network.Classify_node(args).to(dev); m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4, fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        F.nll_loss(m(x, ei).index_select(0, idx), y.index_select(0, idx)).backward()
        opt.step()
    print(f"{layer:9s}: {timeit(step):7.2f} ms per step (model(x, edge_index) through the nn.Module surface, torch Adam)", flush=True)
class APPNPNet(torch.nn.Module):   # Baselines/SGGC/APPNP/networks.py: lin1 -> relu -> dropout -> lin2 -> APPNP(K=10, alpha=0.1)
    def __init__(self):
        super().__init__()
        self.l1, self.l2, self.prop = fnn.Linear(Fdim, 512), fnn.Linear(512, C), fnn.APPNP(10, 0.1)
    def forward(self, x, ei):
        h = F.dropout(F.relu(self.l1(F.dropout(x, 0.5, self.training))), 0.5, self.training)
        return F.log_softmax(self.prop(self.l2(h), ei), dim=1)
m = APPNPNet().to(dev); m.train()
opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    F.nll_loss(m(x, ei).index_select(0, idx), y.index_select(0, idx)).backward()
    opt.step()
print(f"APPNP    : {timeit(step):7.2f} ms per step (K = 10 propagations of the {C}-wide logits, forward + backward)")
