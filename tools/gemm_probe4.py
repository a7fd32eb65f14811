import sys, torch
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
R, H = 90549, 512
dev = "cuda"
dH = torch.randn(R, H, device=dev); X1 = torch.randn(R, H, device=dev)
ref = dH.double().t() @ X1.double()
for prec in ("highest", "high"):
    torch.set_float32_matmul_precision(prec)
    for B in (16, 32, 64, 128, 256):
        Kc = R // B
        main = B * Kc
        def f():
            part = torch.bmm(dH[:main].view(B, Kc, H).transpose(1, 2), X1[:main].view(B, Kc, H))
            out = part.sum(0)
            if main < R:
                out = out + dH[main:].t() @ X1[main:]
            return out
        us = timeit(f)
        err = float((f().double() - ref).abs().max() / ref.abs().max())
        print(f"{prec:8s} bmm split-K B={B:3d} Kc={Kc:5d}: {us:8.1f} us   rel err {err:.2e}")
