"""Round 3: fitgnn_gemm_exact_f32 alone, by shape and forced k-chunk count (FITGNN_GEMM_CHUNKS is read per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import torch
from fitgnn_amd import ops

def bench(form, a, b, reps=5):
    ops.gemm_exact(a, b, form); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm_exact(a, b, form)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

shapes = [("nt", 34493, 512, 8448), ("tn", 512, 8448, 34493), ("nt", 165000, 512, 512), ("nn", 165000, 512, 512), ("tn", 512, 512, 165000),
          ("nt", 34493, 512, 512), ("nt", 19717, 512, 512)]
for form, I, J, K in shapes:
    if form == "nt": a, b = torch.randn(I, K, device="cuda"), torch.randn(J, K, device="cuda")
    elif form == "nn": a, b = torch.randn(I, K, device="cuda"), torch.randn(K, J, device="cuda")
    else: a, b = torch.randn(K, I, device="cuda"), torch.randn(K, J, device="cuda")
    flops = 2.0 * I * J * K
    line = []
    for c in sys.argv[1:] or ["0"]:
        if c == "0": os.environ.pop("FITGNN_GEMM_CHUNKS", None)
        else: os.environ["FITGNN_GEMM_CHUNKS"] = c
        ms = bench(form, a, b)
        line.append(f"c={c}: {ms:.3f} ms {flops / ms / 1e9:.0f} TF")
    print(form, (I, J, K), " | ".join(line), flush=True)
    t0 = time.time(); lib = (a @ b.t()) if form == "nt" else (a @ b) if form == "nn" else (a.t() @ b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): lib = (a @ b.t()) if form == "nt" else (a @ b) if form == "nn" else (a.t() @ b)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    print("     library fp32:", f"{ms:.3f} ms {flops / ms / 1e9:.0f} TF", flush=True)
    del a, b
