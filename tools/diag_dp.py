"""Diagnostic (round 3): a rank's shard assembled alone (assemble_subgraphs_torch(clusters=...)) against the same clusters cut
out of the whole union (select_clusters), and the loss shares of both against the whole union's loss, in ONE process."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import numpy as np, torch
from fitgnn_amd import data, network, ops, workloads
from fitgnn_amd.ops import SoftmaxNLL

name = sys.argv[1] if len(sys.argv) > 1 else "S-pubmed"
dev = torch.device("cuda")
wl = workloads.coarsen_workload(name, dev)
ei_d, assign_d = torch.from_numpy(wl["ei"]).to(dev), torch.from_numpy(wl["assign"]).to(dev)
n_c = wl["n_clusters"]
full, nnz_c = workloads.assemble(name, ei_d, assign_d, n_c)
X, y = workloads.features_and_labels(name)
N, E, F, C, r = workloads.SHAPES[name]
margs = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=512, num_classes=C, dropout=0.0)
torch.manual_seed(2)
model = network.Classify_node(margs).to(dev)
model.train()


def loss_of(sub, scale):
    b = workloads.batch_from_subgraphs(name, sub, dev, X, y)
    z = model.embed_and_head(b.x_table, b.edge_index, b.row_index, loss_rows=b.train_idx, compact_logits=True)
    ar = torch.arange(z.shape[0], device=dev)
    return float(SoftmaxNLL.apply(z, ar, b.y.index_select(0, b.train_idx), scale)), int(b.train_idx.numel()), z.detach()


owner = workloads.shard_before_assembly(name, ei_d, assign_d, n_c, 2)
lf, cnt, zf = loss_of(full, 1.0 / N)
print("full loss", lf, "count", cnt)
tot_a = tot_b = 0.0
for k in range(2):
    mine = np.nonzero(owner == k)[0]
    a, _ = workloads.assemble(name, ei_d, assign_d, n_c, clusters=mine)
    b = data.select_clusters(full, mine)
    same = {key: bool(torch.equal(a[key], b[key])) for key in a}
    la, ca, za = loss_of(a, 1.0 / N)
    lb, cb, zb = loss_of(b, 1.0 / N)
    print("rank", k, "clusters", len(mine), "owner crc", int((owner * (np.arange(len(owner)) % 65521 + 1)).sum()), "rows", int(a["ptr"][-1]), "nnz'", int(data.cluster_nnz(a).sum()))
    print("rank", k, "dict equal:", same, "loss a", la, "loss b", lb, "counts", ca, cb, "logits equal", bool(torch.equal(za, zb)))
    tot_a += la; tot_b += lb
print("sum of shares: assembled alone", tot_a, " cut from the union", tot_b, " full", lf)
