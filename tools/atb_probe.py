"""GPU probe: the hand-written a^T @ b kernel against fp64 and against the library path (mm_at_b)."""
import sys, time
sys.path.insert(0, "fit-gnn_amd")
import torch
from fitgnn_amd import ops

def t_us(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

torch.manual_seed(0)
for (R, M, N) in [(90549, 512, 512), (19717, 512, 500), (90549, 512, 500), (1000, 64, 128), (33, 8, 4), (8245538, 512, 100)]:
    if R * max(M, N) > 3e9: continue
    a = torch.randn(R, M, device="cuda"); b = torch.randn(R, N, device="cuda")
    ref = (a.double().t() @ b.double())
    got = ops.gemm_atb(a, b)
    ops.ATB_KERNEL = False          # mm_at_b then takes the library's batched split-K path
    lib = ops.mm_at_b(a, b)
    den = ref.abs().max()
    print(R, M, N, "err hip %.2e lib %.2e" % (float((got - ref).abs().max() / den), float((lib - ref).abs().max() / den)),
          "hip %.1f us  lib %.1f us" % (t_us(lambda: ops.gemm_atb(a, b)), t_us(lambda: ops.mm_at_b(a, b))), flush=True)
    ops.ATB_KERNEL = True
    g2 = ops.gemm_atb(a, b)
    assert torch.equal(got, g2)
