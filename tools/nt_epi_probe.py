"""GPU probe: GEMM with the fused epilogue backward against mm + fitgnn_epilogue_bwd_f32."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fit-gnn_amd"))
import torch
from fitgnn_amd import ops, _lib

def t_us(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

torch.manual_seed(0)
EPI = _lib.EPI_ELU | _lib.EPI_DROPOUT
for (R, N, K) in [(90549, 512, 512), (1000, 100, 64), (300, 260, 32)]:
    a = torch.randn(R, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    out = torch.randn(R, N, device="cuda")
    for mask in [None, (torch.rand(R, N, device="cuda") > 0.5).to(torch.uint8)]:
        dZ, db = ops.gemm_nt_epilogue_bwd(a, b, out, EPI, p=0.5, seed=1234, mask=mask)
        dZ2, db2 = ops.epilogue_bwd_raw(ops.gemm_nt(a, b), out, EPI, p=0.5, seed=1234, mask=mask)
        print(R, N, K, "mask" if mask is not None else "seed", "dZ equal", bool(torch.equal(dZ, dZ2)),
              "db rel err %.2e" % float((db - db2).abs().max() / db2.abs().max()), flush=True)
    if R > 50000:
        print("fused %.1f us; gemm + epilogue kernel %.1f us" % (
            t_us(lambda: ops.gemm_nt_epilogue_bwd(a, b, out, EPI, p=0.5, seed=1234)),
            t_us(lambda: ops.epilogue_bwd_raw(ops.gemm_nt(a, b), out, EPI, p=0.5, seed=1234))), flush=True)
