#!/usr/bin/env python3
"""GPU-box tool: the aten / library ops of ONE eager S-qm9 batch step (forward, L1 loss, backward, Adam), by name, with the source
lines that issued them -- which small torch ops are left in the launch-bound graph-level step.   python tools/qm9_step_ops.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import ProfilerActivity, profile

from fitgnn_amd import graph_data, network, train

n = 4096
mol = graph_data.synthetic_molecules(n, seed=0)
gset = graph_data.GraphSet(mol, ratio=0.5, extra_node=True, device="cuda")
args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=11, hidden=512, num_classes=1, dropout=0.5)
torch.manual_seed(2)
model = network.Regress_graph_gs(args).cuda()
tr = train.GraphTrainer(model, gset, list(range(n // 2)), kind="gs", batch_size=128, lr=0.001, capture=False)
model.train()
for b in tr.batches[:3]:
    tr._one(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for b in tr.batches[3:7]:
        tr._one(b)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=50, max_src_column_width=90))
