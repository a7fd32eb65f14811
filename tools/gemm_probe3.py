import torch
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
R, H, C = 90549, 512, 3
dev = "cuda"
out = torch.randn(R, H, device=dev); dy = torch.randn(R, C, device=dev); Wl = torch.randn(C, H, device=dev)
print("fwd  out @ Wl^T      ", timeit(lambda: torch.mm(out, Wl.t())))
print("dX   dy @ Wl         ", timeit(lambda: torch.mm(dy, Wl)))
print("dWl  dy^T @ out      ", timeit(lambda: torch.mm(dy.t(), out)))
print("dWl  (out^T @ dy)^T  ", timeit(lambda: torch.mm(out.t(), dy).t()))
print("dWl  via sum (C=3)   ", timeit(lambda: torch.stack([(out * dy[:, c:c+1]).sum(0) for c in range(C)])))
print("dWl  einsum          ", timeit(lambda: torch.einsum('rc,rh->ch', dy, out)))
print("log_softmax+nll fwd/bwd small ops n/a")
