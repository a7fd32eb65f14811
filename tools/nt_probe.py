"""GPU probe: the hand-written a @ b^T kernel against fp64 and against the library ("high" precision torch.mm)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fit-gnn_amd"))
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
for (R, N, K) in [(90549, 512, 512), (300, 512, 512), (1000, 100, 64), (19717, 512, 480), (5, 4, 32), (489540, 512, 512)]:
    a = torch.randn(R, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    ref = a.double() @ b.double().t()
    got = ops.gemm_nt(a, b)
    lib = ops.mm(a, b.t())   # torch.mm under the "high" precision policy: the library's 3 x bf16 kernel
    den = ref.abs().max()
    print(R, N, K, "err hip %.2e lib %.2e" % (float((got - ref).abs().max() / den), float((lib - ref).abs().max() / den)),
          "hip %.1f us  lib %.1f us" % (t_us(lambda: ops.gemm_nt(a, b)), t_us(lambda: ops.mm(a, b.t()))), flush=True)
    assert torch.equal(got, ops.gemm_nt(a, b))
