#!/usr/bin/env python3
"""GPU-box tool: A/B two builds of the SpMM entry point (libfitgnn_hip.so vs another .so) on the S-pubmed union,
interleaved rounds in one process."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from fitgnn_amd import _lib
from fitgnn_amd.csr import CSRGraph

other = sys.argv[1]
windows = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [12, 16]
dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
R = batch.n_rows
libs = {"new": _lib.lib(), "old": ctypes.CDLL(other)}
res, args = _lib.SIGNATURES["fitgnn_spmm_csr_f32"]
libs["old"].fitgnn_spmm_csr_f32.restype, libs["old"].fitgnn_spmm_csr_f32.argtypes = res, args
X = torch.randn(R, 512, device=dev); Y = torch.empty_like(X)
def timeit(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for w in windows:
    g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr, lds_rows=w)
    f = g.f
    out = {k: [] for k in libs}
    for rnd in range(6):
        for k, L in libs.items():
            fn = lambda: L.fitgnn_spmm_csr_f32(_lib.dptr(f.rowptr), _lib.dptr(f.col), _lib.dptr(f.val), _lib.dptr(X), 512, _lib.dptr(Y), 512,
                                               R, 512, _lib.dptr(f.tiles), int(f.tiles.shape[0]), None, None, None, w, None, 0, 0.0, 0, None,
                                               _lib.stream_ptr(dev))
            out[k].append(timeit(fn))
    print(f"window {w}: " + "  ".join(f"{k} median {np.median(v):.1f} min {min(v):.1f}" for k, v in out.items()))
