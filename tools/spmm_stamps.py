#!/usr/bin/env python3
"""GPU-box tool: cycle breakdown of a workgroup of the whole-subgraph SpMM kernel on the S-products union, for the three launch
kinds of the step (plain; operand table behind a row indirection; compact operand + the previous layer's derivative in the store).
Needs the counters:  touch fit-gnn_amd/csrc/spmm.hip && make -C fit-gnn_amd/csrc EXTRA=-DFITGNN_SPMM_STAMPS  (rebuild without afterwards;
add -DFITGNN_SPMM_NOSTORE to compile the row stores out: what the kernel takes when nothing is written)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fitgnn_amd import _lib, ops, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "S-products"
dev = torch.device("cuda:0")
co = workloads.coarsen_workload(name, dev)
sub, _ = workloads.assemble(name, torch.from_numpy(co["ei"]).to(dev), torch.from_numpy(co["assign"]).to(dev), co["n_clusters"])
b = workloads.batch_from_subgraphs(name, sub, dev)
g, H = b.graph, 512
N = workloads.SHAPES[name][0]
L = _lib.lib()
L.fitgnn_debug_spmm_buffer.argtypes = [ctypes.c_void_p]
n_wg = (int(g.f.blocks.shape[0]) + 7) // 8 * 8 * 2 + 64
dbg = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
L.fitgnn_debug_spmm_buffer(ctypes.c_void_p(dbg.data_ptr()))
X = torch.randn(g.n, H, device=dev)
table = torch.randn(N, H, device=dev)
rows = b.train_idx
comp = torch.cat([torch.randn(rows.numel(), H, device=dev), torch.zeros(ops.ZERO_ROWS, H, device=dev)])
pos = ops._compact_positions(g, rows)
prev = torch.randn(g.n, H, device=dev)
kinds = {
    "plain (layer 1 forward)": lambda: ops.spmm_graph(g, X),
    "plain transposed (layer 0 backward)": lambda: ops.spmm_graph(g, X, transposed=True),
    "operand table (layer 0 forward)": lambda: ops.spmm_graph(g, table, xrow=b.row_index.index),
    "compact operand (zero rows from LDS)": lambda: ops.spmm_graph(g, comp, transposed=True, xrow=pos, zero_from=int(rows.numel())),
    "compact operand + derivative in the store": lambda: ops.spmm_graph_dz(g, comp, prev, _lib.EPI_ELU | _lib.EPI_DROPOUT, p=0.5, seed=7, xrow=pos,
                                                                           zero_from=int(rows.numel())),
}
link = ops.EpilogueLink()
link.record(True, 0.5, 7, None, True, g=g)
if g.t.blocks is not None:
    kinds["two-hop backward (dZ made in the window)"] = lambda: ops.spmm_two_hop_blocks(g, comp, prev, rows, pos, link)
names = ["start-up (record -> first barrier)", "publish + barrier (waits for the prefetch)", "short rows", "long rows", "end-of-piece barrier", "tail (long rows out)"]
for kind, fn in kinds.items():
    fn(); torch.cuda.synchronize()
    dbg.zero_(); torch.cuda.synchronize()
    t0 = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t0
    d = dbg.view(-1, 8)
    live = d[:, 6] > 0
    v = d[live].sum(0).tolist() + [int(live.sum())]
    n = max(v[8], 1)
    print(f"{kind}: {dt*1e3:.2f} ms, {v[8]} workgroups, {v[7]/n:.2f} pieces each, lifetime {v[6]/n:.0f} cycles")
    for nm, c in zip(names, v[:6]):
        print(f"    {nm:46s} {100.0*c/max(v[6],1):5.1f} %   {c/n:8.0f} cycles per workgroup")
