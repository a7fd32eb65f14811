#!/usr/bin/env python3
"""GPU-box tool: where the subgraph assembly (f1) and the batch-CSR build of a workload spend their time -- selected functions wrapped
with a device synchronisation and a wall-clock timer.   python tools/time_assembly.py [S-products]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch

from fitgnn_amd import csr, data, ops, workloads

TIMES = {}


def timed(owner, name):
    fn = getattr(owner, name)

    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.time()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        TIMES.setdefault(f"{getattr(owner, '__name__', owner.__class__.__name__)}.{name}", []).append(time.time() - t0)
        return out
    setattr(owner, name, wrap)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "S-products"
    dev = torch.device("cuda")
    w0 = workloads.coarsen_workload(wl, dev)
    ei_d = torch.from_numpy(w0["ei"]).to(dev)
    assign_d = torch.from_numpy(w0["assign"]).to(dev)
    for owner, names in ((data, ["assemble_subgraphs_torch", "cluster_nnz"]),
                         (csr, ["_csr_from_coo", "make_tiles", "split_blocks", "tiles_to_device", "stream_ranges", "block_boundaries",
                                "arrange_tiles_for_xcds"]),
                         (csr.CSRGraph, ["finalize", "__init__"]), (ops.RowIndex, ["__init__"]), (data.SubgraphBatch, ["__init__"])):
        for n in names:
            timed(owner, n)
    if "--prewarm" in sys.argv:   # the same torch / library kernels on a small graph first: is the first call's cost code loading or allocation?
        w1 = workloads.coarsen_workload("S-pubmed", dev)
        e1, a1 = torch.from_numpy(w1["ei"]).to(dev), torch.from_numpy(w1["assign"]).to(dev)
        torch.cuda.synchronize(); t0 = time.time()
        s1, _ = workloads.assemble("S-pubmed", e1, a1, w1["n_clusters"])
        b1 = workloads.batch_from_subgraphs("S-pubmed", s1, dev)
        torch.cuda.synchronize()
        print(f"prewarm on S-pubmed: {time.time() - t0:.3f} s")
        del s1, b1
    for rnd in range(2):
        TIMES.clear()
        torch.cuda.synchronize(); t0 = time.time()
        sub, _ = workloads.assemble(wl, ei_d, assign_d, w0["n_clusters"])
        torch.cuda.synchronize(); t1 = time.time()
        batch = workloads.batch_from_subgraphs(wl, sub, dev)
        torch.cuda.synchronize(); t2 = time.time()
        print(f"round {rnd}: assemble {t1 - t0:.3f} s, batch {t2 - t1:.3f} s (rows {batch.n_rows}, nnz' {batch.nnz})")
        for k, v in sorted(TIMES.items(), key=lambda kv: -sum(kv[1])):
            print(f"   {k:45s} x{len(v)}  {sum(v) * 1e3:9.1f} ms")
        del batch, sub


if __name__ == "__main__":
    main()
