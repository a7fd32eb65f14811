#!/usr/bin/env python3
"""GPU-box tool: would layer 0's table launch gain from COLUMN-SLAB passes narrow enough for the slab of the table to stay in the 256-MiB
Infinity Cache?  The whole-subgraph kernel is launched once per slab of W columns (X / Y pointers offset, leading dimensions kept,
H = W: lanes beyond W idle -- an inefficient stand-in for a kernel built for narrow slabs) over (a) the real de-duplicated table through
the row indirection and (b) a SEQUENTIAL [R, H] operand through an identity indirection (no row is ever re-read: nothing a cache can
keep).  (a) much faster than (b) at small W = the table slab is served on-die.   python tools/table_slab_probe.py [S-products]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import ctypes

import torch

from fitgnn_amd import _lib, ops, workloads


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
    dev = torch.device("cuda")
    w0 = workloads.coarsen_workload(wl, dev)
    sub, _ = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
    batch = workloads.batch_from_subgraphs(wl, sub, dev)
    g, H = batch.graph, 512
    side = g.f
    R, n_table = g.n, int(batch.x_table.shape[0])
    xrow = batch.row_index.index
    T = torch.randn(n_table, H, device=dev)
    Tseq = torch.randn(R, H, device=dev)
    ar = torch.arange(R, dtype=torch.int32, device=dev)
    Y = torch.empty((R, H), dtype=torch.float32, device=dev)
    L = _lib.lib()
    st = _lib.stream_ptr(dev)
    xcol_t = ops._entry_rows(side, xrow)
    xcol_s = side.col.clone()
    side.xcol = None

    def launch(X, xr, xc, W):
        for c0 in range(0, H, W):
            xp = ctypes.c_void_p(X.data_ptr() + 4 * c0)
            yp = ctypes.c_void_p(Y.data_ptr() + 4 * c0)
            _lib.check(L.fitgnn_spmm_csr_blocks_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), xp, H, yp, H, R, W,
                                                    _lib.dptr(side.blocks), int(side.blocks.shape[0]), _lib.dptr(side.long_rows), _lib.dptr(xr),
                                                    _lib.dptr(xc), -1, None, 0, 0.0, 0, None, st), "blocks")

    def timeit(fn, n=4):
        fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3

    print(f"{wl}: rows {R}, table rows {n_table} ({n_table * H * 4 / 2**20:.0f} MiB), nnz' {int(side.col.numel())}", flush=True)
    for W in (512, 256, 128, 64):
        a = min(timeit(lambda: launch(T, xrow, xcol_t, W)) for _ in range(2))
        b = min(timeit(lambda: launch(Tseq, ar, xcol_s, W)) for _ in range(2))
        print(f"  slab {W:3d} columns ({n_table * W * 4 / 2**20:5.0f} MiB of table per pass, {H // W} passes): table {a:9.1f} us   sequential {b:9.1f} us   ratio {a / b:.3f}",
              flush=True)


if __name__ == "__main__":
    main()
