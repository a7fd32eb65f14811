#!/usr/bin/env python3
"""GPU-box tool for rocprofv3: a few GATConv training steps on the S-pubmed union (arg: layer name)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import bench
from fitgnn_amd import network
layer = sys.argv[1] if len(sys.argv) > 1 else "GATConv"
dev = torch.device("cuda")
batch, (Fdim, C), info = bench.build_workload("S-pubmed", 0, dev)
args = argparse.Namespace(num_layers1=2, layer_name=layer, num_features=Fdim, hidden=512, num_classes=C)
m = network.Classify_node(args).to(dev); m.train()
opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4, fused=True)
for _ in range(8):
    opt.zero_grad(set_to_none=True)
    F.nll_loss(m(batch.x, batch.edge_index).index_select(0, batch.train_idx), batch.y.index_select(0, batch.train_idx)).backward()
    opt.step()
torch.cuda.synchronize()
