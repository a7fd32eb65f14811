import torch, time
x = torch.empty(4 << 30, dtype=torch.float32, device="cuda")  # 16 GiB
for name, fn in [("zero_", lambda: x.zero_()), ("fill_(1.5)", lambda: x.fill_(1.5)), ("copy half->half", lambda: x[: 2 << 30].copy_(x[2 << 30:]))]:
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 5
    print(f"{name}: {dt*1e3:.2f} ms, {x.numel()*4/dt/1e12:.2f} TB/s written" if "copy" not in name else f"{name}: {dt*1e3:.2f} ms, {x.numel()*4/dt/1e12:.2f} TB/s read+written")
