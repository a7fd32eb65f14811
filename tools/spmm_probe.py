#!/usr/bin/env python3
"""GPU-box tool: time the SpMM variants on a workload's union (interleaved rounds in one process, HIP events), next to a
device copy of the same byte count:  python tools/spmm_probe.py S-products 16,24,32"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from fitgnn_amd import ops, workloads
from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU, SPMM_GATHER
from fitgnn_amd.csr import CSRGraph


def timeit(fn, n=10):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
    windows = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 24, 32]
    variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["contig", "planned", "gather"]
    dev = torch.device("cuda")
    w0 = workloads.coarsen_workload(wl, dev)
    sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
    layout = os.environ.get("PROBE_LAYOUT", "star")
    if layout != "star":
        sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"], layout=layout)
    ptr = sub["ptr"].cpu().numpy()
    R, nnz = int(ptr[-1]), int(nnz_c.sum())
    if "seg_start" in sub:   # the kernels' units: the stars
        ptr = np.concatenate([np.nonzero(sub["seg_start"].cpu().numpy())[0], [R]]).astype(np.int64)
    H = 512
    X = torch.randn(R, H, device=dev)
    Y = torch.empty_like(X)
    b = torch.randn(H, device=dev)
    bytes_spmm = 8 * H * R + 8 * nnz + 4 * (R + 1)
    sizes = np.diff(ptr)
    print(f"{wl}: rows {R} nnz' {nnz} ({nnz / R:.2f}/row) subgraphs {len(sizes)} rows/subgraph mean {sizes.mean():.1f} max {sizes.max()} "
          f"bytes/launch {bytes_spmm / 1e6:.1f} MB", flush=True)
    if "stats" in variants:   # how much of the pattern leaves its segment (what the whole-subgraph kernel still has to gather)
        e = sub["edge_index"]
        seg_of = torch.from_numpy(np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))).to(dev)
        deg = torch.bincount(e[1], minlength=R) + 1
        out = seg_of[e[0]] != seg_of[e[1]]
        long_t = deg[e[1]] > 16
        segsz = torch.from_numpy(np.diff(ptr)).to(dev)
        big = segsz[seg_of[e[1]]] > 16
        print(f"entries leaving their segment: {int(out.sum())} of {e.shape[1]} ({float(out.float().mean()):.3f}); in long rows {int((out & long_t).sum())}, "
              f"in short rows {int((out & ~long_t).sum())}; in segments > 16 rows {int((out & big).sum())}; long rows {int((deg > 16).sum())}; "
              f"rows in segments > 16 rows with > 4 long rows: {int(segsz[torch.bincount(seg_of[deg > 16], minlength=len(ptr) - 1) > 4].sum())}", flush=True)
    graphs = {}
    if "nohub" in variants:   # the same union with the entries of long rows (> 32 non-zeros: the hubs) removed: what their tails cost
        e = sub["edge_index"]
        deg = torch.bincount(e[1], minlength=R)
        keep = deg[e[1]] <= 32
        graphs[("nohub", windows[0])] = CSRGraph(e[:, keep].contiguous(), R, mode="gcn", ptr=ptr, lds_rows=windows[0], planned=False)
        print(f"nohub: dropped {int((~keep).sum())} of {e.shape[1]} edges ({int((deg > 32).sum())} rows longer than 32)", flush=True)
    if "nomiss" in variants:  # only the entries whose operand row sits in the output row's own tile: every entry is a window hit
        from fitgnn_amd.csr import make_tiles
        tl = make_tiles(ptr, windows[0])
        tile_of = torch.from_numpy(np.repeat(np.arange(len(tl)), tl["row_end"] - tl["row_begin"])).to(dev)
        e = sub["edge_index"]
        keep = tile_of[e[0]] == tile_of[e[1]]
        graphs[("nomiss", windows[0])] = CSRGraph(e[:, keep].contiguous(), R, mode="gcn", ptr=ptr, lds_rows=windows[0], planned=False)
        print(f"nomiss: dropped {int((~keep).sum())} of {e.shape[1]} edges", flush=True)
    for w in windows:
        if "contig" in variants:
            graphs[("contig", w)] = CSRGraph(sub["edge_index"], R, mode="gcn", ptr=ptr, lds_rows=w, planned=False)
        if "planned" in variants:
            graphs[("planned", w)] = CSRGraph(sub["edge_index"], R, mode="gcn", ptr=ptr, lds_rows=w, planned=True)
    if "gather" in variants:
        graphs[("gather", 0)] = CSRGraph(sub["edge_index"], R, mode="gcn", ptr=ptr, planned=False, gather=True)
    # A/B bit of the LDS-window kernel: one tile per workgroup (round 1) instead of the pipelined form
    flagsets = {"": 0}
    if "ab" in variants:
        flagsets.update({"tiled": "tiled"})
        for k in list(graphs):
            if k[0] == "contig":
                for nm in flagsets:
                    if nm:
                        graphs[(k[0] + ":" + nm, k[1])] = graphs[k]
    res = {k: {"plain": [], "epi": []} for k in graphs}
    copy = []
    for rnd in range(4):
        for k, g in graphs.items():
            fl = flagsets.get(k[0].partition(":")[2], 0)
            cfg = ops.OpConfig(split_large_blocks=fl != "tiled")   # "tiled": large blocks cut into window-sized tiles (round 1)
            fl = 0 if fl == "tiled" else fl
            res[k]["plain"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y, epilogue=fl, cfg=cfg)))
            res[k]["epi"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y, bias=b, epilogue=fl | EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=7, cfg=cfg)))
        copy.append(timeit(lambda: Y.copy_(X)))
    print(f"copy of {8 * H * R / 1e6:.1f} MB: {np.median(copy):.1f} us = {8 * H * R / np.median(copy) / 1e3:.0f} GB/s")
    for k, g in graphs.items():
        p, e = np.median(res[k]["plain"]), np.median(res[k]["epi"])
        miss = float((g.f.lcol < 0).float().mean()) if g.f.lcol is not None else float("nan")
        print(f"{k[0]:24s} window {k[1]:3d} tiles {g.f.n_tiles:7d} planner-miss {miss:.3f}: plain {p:9.1f} us ({bytes_spmm / p / 1e3:6.0f} GB/s, "
              f"{bytes_spmm / p / 1e3 / 8000:.3f} of 8 TB/s)  epilogue {e:9.1f} us ({bytes_spmm / e / 1e3:6.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
