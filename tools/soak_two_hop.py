#!/usr/bin/env python3
"""GPU-box tool: the two-hop backward against the two separate launches on random star batches (sizes, centres per star, leaf -- leaf
density, H, which rows reach the loss, dropout on / off / masked): A^T dZ bit for bit, column sums to summation order.
python tools/soak_two_hop.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from fitgnn_amd import csr, ops
from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU


def star_blocks(rng, sizes, centres, extra):
    src, dst, off = [], [], 0
    for s in sizes:
        c = min(centres, max(s - 1, 0))
        for h in range(c):
            leaves = np.arange(c, s)
            src += [off + h] * len(leaves) + (off + leaves).tolist()
            dst += (off + leaves).tolist() + [off + h] * len(leaves)
        m = int(extra * s * s)
        if m and s > 2:
            a, b = rng.integers(0, s, size=m), rng.integers(0, s, size=m)
            k = a != b
            src += (off + a[k]).tolist() + (off + b[k]).tolist()
            dst += (off + b[k]).tolist() + (off + a[k]).tolist()
        off += s
    if not src:
        src, dst = [0], [0]
    return torch.tensor(np.unique(np.array([src, dst]), axis=1), dtype=torch.long), off


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = ran = 0
    for case in range(cases):
        nb = int(rng.integers(1, 30))
        sizes = [int(x) for x in rng.choice([1, 2, 3, 5, 16, 17, 18, 31, 33, 64, 65, 100, 257, 700], size=nb)]
        centres = int(rng.integers(1, 6))
        extra = float(rng.choice([0.0, 0.002, 0.02, 0.1]))
        H = int(rng.choice([4, 64, 96, 256, 260, 512]))
        ei, n = star_blocks(rng, sizes, centres, extra)
        ptr = np.concatenate([[0], np.cumsum(sizes)])
        limit = int(rng.choice([4096, 64, 256]))
        g = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=limit)
        if g.t.blocks is None:
            continue
        mode = rng.choice(["centres", "first", "random", "all", "one"])
        if mode == "centres":
            rows = np.concatenate([np.arange(min(centres, max(s - 1, 1))) + o for s, o in zip(sizes, ptr[:-1])])
        elif mode == "first":
            rows = ptr[:-1].copy()
        elif mode == "random":
            rows = np.sort(rng.choice(n, size=max(1, n // int(rng.integers(2, 20))), replace=False))
        elif mode == "all":
            rows = np.arange(n)
        else:
            rows = np.array([int(rng.integers(0, n))])
        rows = torch.from_numpy(np.unique(rows).astype(np.int64)).cuda()
        k = int(rows.numel())
        Xc = torch.cat([torch.randn(k, H).cuda(), torch.zeros(ops.ZERO_ROWS, H).cuda()])
        prev = torch.randn(n, H).cuda() * (torch.rand(n, H).cuda() > 0.3)
        pos = ops._compact_positions(g, rows)
        which = int(rng.integers(0, 3))
        flags, p, mask = [(EPI_ELU | EPI_DROPOUT, 0.5, None), (EPI_ELU | EPI_DROPOUT, 0.3, (torch.rand(n, H).cuda() > 0.3).to(torch.uint8)),
                          (EPI_ELU, 0.0, None)][which]
        link = ops.EpilogueLink()
        link.record(bool(flags & EPI_DROPOUT), p, 1000 + case, mask, True, g=g)
        G, db = ops.spmm_two_hop_blocks(g, Xc, prev, rows, pos, link)
        dZ, want_db = ops.spmm_graph_dz(g, Xc, prev, flags, p=p, seed=1000 + case, mask=mask, want_db=True, xrow=pos, zero_from=k)
        want = ops.spmm_graph(g, dZ, transposed=True)
        ran += 1
        ok = torch.equal(G, want)
        scale = float(want_db.abs().max())
        ok_db = scale == 0.0 or float((db - want_db).abs().max()) <= 2e-5 * scale
        if not (ok and ok_db):
            bad += 1
            print("MISMATCH case", case, dict(sizes=sizes, centres=centres, extra=extra, H=H, limit=limit, mode=str(mode), which=which, G=ok, db=ok_db), flush=True)
    print("cases run:", ran, "mismatches:", bad)


if __name__ == "__main__":
    main()
