#!/usr/bin/env python3
"""GPU-box tool: fitgnn_gemm_exact_f32 by tile shape (FITGNN_GEMM_SHAPE is read per call) on the short operands of the configurations,
against the default plan and the library's fp32 GEMM.  python tools/gemm_shape_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import torch
from fitgnn_amd import ops


def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [("nt", 34493, 512, 8448), ("tn", 512, 8448, 34493), ("nt", 4861, 512, 512), ("nn", 4861, 512, 512), ("tn", 512, 512, 4861), ("nt", 19717, 512, 512), ("nn", 19717, 512, 512),
          ("tn", 512, 512, 19717), ("nt", 20625, 512, 512), ("nt", 34493, 512, 512), ("nt", 90549, 512, 512), ("nt", 165000, 512, 128),
          ("tn", 512, 128, 165000), ("tn", 512, 512, 165000), ("tn", 512, 512, 20625), ("nt", 165000, 512, 512), ("nt", 19717, 512, 512 - 12), ("nt", 82500, 48, 512)]
names = {"0": "256x256", "3": "128x128", "4": "64x128", "6": "64x64"}
for form, I, J, K in shapes:
    if form == "nt": a, b = torch.randn(I, K, device="cuda"), torch.randn(J, K, device="cuda")
    elif form == "nn": a, b = torch.randn(I, K, device="cuda"), torch.randn(K, J, device="cuda")
    else: a, b = torch.randn(K, I, device="cuda"), torch.randn(K, J, device="cuda")
    flops = 2.0 * I * J * K
    os.environ.pop("FITGNN_GEMM_SHAPE", None)
    us = bench(lambda: ops.gemm_exact(a, b, form))
    line = [f"default {us:8.1f} us {flops / us / 1e6:6.1f} TF"]
    os.environ["FITGNN_GEMM_DEEP"] = "1"   # the small shapes with a two-stage look-ahead (measured slower: opt-in)
    us = bench(lambda: ops.gemm_exact(a, b, form))
    os.environ.pop("FITGNN_GEMM_DEEP", None)
    line.append(f"default, two-stage look-ahead {us:8.1f} us {flops / us / 1e6:6.1f} TF")
    for sh in ("0", "3", "4", "6"):
        os.environ["FITGNN_GEMM_SHAPE"] = sh
        us = bench(lambda: ops.gemm_exact(a, b, form))
        line.append(f"{names[sh]} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
        if K >= 4096 and form != "tn":   # a long k: the same shape without the k split
            os.environ["FITGNN_GEMM_CHUNKS"] = "1"
            us = bench(lambda: ops.gemm_exact(a, b, form))
            line.append(f"{names[sh]} unsplit {us:8.1f} us {flops / us / 1e6:6.1f} TF")
            os.environ.pop("FITGNN_GEMM_CHUNKS", None)
    os.environ.pop("FITGNN_GEMM_SHAPE", None)
    lib = (lambda: a @ b.t()) if form == "nt" else (lambda: a @ b) if form == "nn" else (lambda: a.t() @ b)
    us = bench(lib)
    line.append(f"library {us:8.1f} us {flops / us / 1e6:6.1f} TF")
    print(form, (I, J, K), " | ".join(line), flush=True)
    del a, b
