"""A/B: one GD step over the S-products union as K independent parts (whole subgraphs, data.shard_clusters) issued on K HIP
streams of ONE process, against the single-stream step.  Gradients are additive over subgraphs (run.py:184-204)."""
import argparse
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import faulthandler
import numpy as np, torch
faulthandler.dump_traceback_later(int(os.environ.get("PROBE_DUMP_S", "90")), exit=True)
from fitgnn_amd import data, network, ops, train, workloads
from fitgnn_amd.ops import SoftmaxNLL

name = sys.argv[1] if len(sys.argv) > 1 else "S-products"
Ks = [int(k) for k in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "3", "4"])]
dev = torch.device("cuda")
wl = workloads.coarsen_workload(name, dev)
sub, nnz_c = workloads.assemble(name, torch.from_numpy(wl["ei"]).to(dev), torch.from_numpy(wl["assign"]).to(dev), wl["n_clusters"])
X, y = workloads.features_and_labels(name)
N, E, F, C, r = workloads.SHAPES[name]
margs = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=512, num_classes=C)
for K in Ks:
    torch.manual_seed(2)
    model = network.Classify_node(margs).to(dev)
    flat = train.FlatGrads(model.parameters()); opt = train.FlatAdam(flat)
    owner = data.shard_clusters(None, nnz_c, K)
    parts = [workloads.batch_from_subgraphs(name, data.select_clusters(sub, np.nonzero(owner == k)[0]) if K > 1 else sub, dev, X, y) for k in range(K)]
    ytr = [b.y.index_select(0, b.train_idx) for b in parts]
    count = sum(int(b.train_idx.numel()) for b in parts)
    streams = [torch.cuda.Stream() for _ in range(K)]
    cur = torch.cuda.current_stream()
    model.train()

    def step():
        flat.zero()
        losses = []
        for b, yt, s in zip(parts, ytr, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                z = model.embed_and_head(b.x_table, b.edge_index, b.row_index, loss_rows=b.train_idx)
                losses.append(SoftmaxNLL.apply(z, b.train_idx, yt, 1.0 / count))
        torch.autograd.backward(losses)
        for s in streams:
            cur.wait_stream(s)
        opt.step()
        return losses

    print(f"K={K}: built", flush=True)
    for i in range(3):
        step()
        torch.cuda.synchronize()
        print(f"K={K}: warm-up step {i} done, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ls = step()
        if os.environ.get("PROBE_SYNC", "1") == "1":
            torch.cuda.synchronize()   # the host never runs more than one step ahead
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"K={K} streams: {dt*1e3:.2f} ms/step  loss={sum(float(l) for l in ls):.6f}", flush=True)
    del parts, model, flat, opt, ls
    torch.cuda.empty_cache()
