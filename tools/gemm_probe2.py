import sys, torch
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
R, F, H, N0 = 90549, 500, 512, 19717
dev = "cuda"
torch.set_float32_matmul_precision(sys.argv[1])
X = torch.randn(R, F, device=dev); X1 = torch.randn(R, H, device=dev); dH = torch.randn(R, H, device=dev)
X0 = torch.randn(N0, F, device=dev); W0 = torch.randn(H, F, device=dev); idx = torch.randint(0, N0, (R,), device=dev)
dHt = dH.t().contiguous(); Xt = X.t().contiguous()
ref = (dH.double().t() @ X1.double())
tests = {
  "dW  dH.t() @ X1 (view)": lambda: torch.mm(dH.t(), X1),
  "dW  dH.t().contiguous() @ X1": lambda: torch.mm(dH.t().contiguous(), X1),
  "dW  (X1.t().contiguous() @ dH).t()": lambda: torch.mm(X1.t().contiguous(), dH),
  "dW  pre-transposed dHt @ X1": lambda: torch.mm(dHt, X1),
  "transpose copy only": lambda: dH.t().contiguous(),
  "dedup fwd0: X0[N0,500] @ W0^T": lambda: torch.mm(X0, W0.t()),
  "dedup gather rows [R,512]": lambda: torch.mm(X0, W0.t())[idx],
  "dedup bwd: index_add + small GEMM": lambda: torch.mm(torch.zeros(N0, H, device=dev).index_add_(0, idx, dH).t(), X0),
  "einsum chunked dW (8 chunks)": lambda: sum(torch.mm(dH[i::8].t(), X1[i::8]) for i in range(8)),
}
for k, fn in tests.items():
    print(f"{k:40s} {timeit(fn):8.1f} us")
err = (torch.mm(dHt, X1).double() - ref).abs().max() / ref.abs().max()
print("rel err dW:", float(err))
