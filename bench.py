#!/usr/bin/env python3
"""bench.py -- edges aggregated/sec (GCN fwd+bwd) on coarsened subgraphs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (config.workload): S-pubmed = BASELINE.json configs[1] "PubMed node_cls FIT-GNN,
variation_neighborhoods r=0.5" on synthetic data of PubMed's shape: preferential-
attachment graph N=19717, E=44324 (graph seed = rank), features U[0,1) row-L1-normalised F=500, 3 classes,
coarsened by the HIP contraction step (r=0.5), one 1-hop "extra node" subgraph per cluster, all
subgraph batches merged into one device-resident block-diagonal CSR.  A step = one GD training epoch of
run.py:177-215: forward over every subgraph (2-layer GCN, hidden 512), one NLL loss, backward, Adam step
(4 SpMM launches; edges aggregated = 4 * nnz').  N>1 = data parallel over subgraph shards: every rank
holds its own S-pubmed-sized shard (weak scaling), one flat RCCL gradient all-reduce per step.

Prints ONE JSON line (rank 0) with `roofline` for the SpMM kernel (HIP-event timed inside the timed
region) and `cpu_baseline` (the torch-CPU oracle of the same step, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling is ~6300

WORKLOADS = {
    # name: (N, E, F, classes, Loukas r)
    "S-pubmed": (19717, 44324, 500, 3, 0.5),
    "S-cora": (2708, 5278, 1433, 7, 0.5),
    "S-physics": (34493, 247962, 8415, 5, 0.7),     # CLI --coarsening_ratio 0.3 (main.py:278 passes 1 - ratio)
    "S-products": (165000, 4125000, 100, 47, 0.5),  # one ogbn-products community (<= 165 000 nodes, main.py:264), mean degree 50 assumed
}


def build_workload(name, seed, device, hidden=512):
    from fitgnn_amd import coarsening, data

    N, E, F, C, r = WORKLOADS[name]
    t0 = time.time()
    ei = data.synthetic_graph(N, E, seed=seed)
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
    G = coarsening.Graph(W)
    # spectral input of the contraction step (host prelude, coarsening_utils.py:83-90), deterministic start vector
    import scipy.sparse.linalg as spla
    offset = 2 * max(G.dw)
    T = offset * sp.eye(N, format="csc") - G.L
    lk, Uk = spla.eigsh(T, k=10, which="LM", tol=1e-5, v0=np.random.default_rng(seed).standard_normal(N))
    lk, Uk = (offset - lk)[::-1], Uk[:, ::-1]
    t1 = time.time()
    torch.cuda.synchronize()
    Cmat, Gc, _ = coarsening.coarsen(G, r=r, method="variation_neighborhoods", Uk=np.ascontiguousarray(Uk), lk=lk.copy(),
                                     device=device)
    torch.cuda.synchronize()
    t2 = time.time()
    assign = sp.csc_matrix(Cmat).indices
    sub = data.assemble_subgraphs_torch(torch.from_numpy(ei).to(device), N, assign, Cmat.shape[0], extra_node=True)
    torch.cuda.synchronize()
    t3 = time.time()
    rng = np.random.default_rng(seed + 1)
    X = rng.random((N, F), dtype=np.float32)
    X /= X.sum(1, keepdims=True)  # --normalize_features (main.py:48)
    y = rng.integers(0, C, size=N)
    train_mask = np.ones(N, dtype=bool)  # every cluster node labelled: every subgraph takes part in the GD step
    batch = data.SubgraphBatch(sub, X, y, train_mask, device=device)
    info = dict(nodes=N, undirected_edges=E, features=F, classes=C, clusters=int(Cmat.shape[0]),
                union_rows=batch.n_rows, nnz_prime=batch.nnz, t_graph_eig_s=round(t1 - t0, 2),
                t_coarsen_hip_s=round(t2 - t1, 3), t_assemble_s=round(t3 - t2, 2), t_batch_csr_s=round(time.time() - t3, 2))
    info["_coarsen_inputs"] = (W, np.ascontiguousarray(Uk), lk.copy(), r)
    return batch, (F, C), info


def cpu_baseline(batch, sd, num_layers, budget_s=20.0):
    """The torch-CPU oracle of the same step (fwd + loss + bwd), timed on this host.  Sample = as many of the
    reference's 128-subgraph loader batches (run.py:336) as fit the time budget, at least 8."""
    from oracle import gnn_oracle as gorc

    x, ei, y = batch.x.cpu(), batch.edge_index.cpu(), batch.y.cpu()
    tm = batch.train_mask.cpu()
    spans = batch.slice_batches(128)

    def run(k):
        r1 = spans[k - 1][1]
        keep = ei[0] < r1
        e = ei[:, keep]
        t0 = time.time()
        gorc.classify_node_fwd_bwd(sd, x[:r1], e, y[:r1], num_layers=num_layers, train_mask=tm[:r1])
        return time.time() - t0, 4 * (int(e.shape[1]) + r1)

    run(min(2, len(spans)))  # warm-up (thread pools, allocator)
    k = min(8, len(spans))
    dt, edges = run(k)
    if dt < budget_s / 4 and k < len(spans):
        k = min(len(spans), max(k + 1, int(k * (budget_s / 2) / max(dt, 1e-3))))
        dt, edges = run(k)
    return dict(value=edges / dt, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                sample=f"first {k} of {len(spans)} loader batches (128 subgraphs each), 1 fwd+bwd step, {dt:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.25 s of timed region: the clock and the caches have settled
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="S-pubmed")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fold", action="store_true", help="A/B: epilogue backward folded into the transposed SpMM (slower, see DESIGN.md)")
    ap.add_argument("--prune-unused-rows", action="store_true",
                    help="NOT the headline configuration: last layer only on the clusters' own nodes (the extra nodes' outputs "
                         "never reach the loss); the edge count of the metric is then the number actually aggregated")
    ap.add_argument("--no-dedup", action="store_true",
                    help="run layer 0's X@W^T on the materialised union rows (one copy per subgraph membership) instead of "
                         "the de-duplicated feature table")
    ap.add_argument("--gemm-precision", default="high", choices=["high", "highest"],
                    help="dense GEMM policy (fitgnn_amd.ops.GEMM_PRECISION): high = fp32 via 3xbf16 split on the forward/dX "
                         "products (rel err ~5e-6), highest = plain fp32 MFMA everywhere")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    n_dev = torch.cuda.device_count()
    local_rank %= max(n_dev, 1)   # rehearsal on a box with fewer GPUs than ranks (FITGNN_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FITGNN_BENCH_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only to rehearse the
        if backend == "nccl":                                       # multi-process path on a single-GPU box
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from fitgnn_amd import network, ops, train

    ops.GEMM_PRECISION = args.gemm_precision
    ops.FOLD_BACKWARD = args.fold

    batch, (F, C), info = build_workload(args.workload, seed=rank, device=device, hidden=args.hidden)
    coarsen_inputs = info.pop("_coarsen_inputs")
    margs = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=F, hidden=args.hidden, num_classes=C)
    torch.manual_seed(2)  # weight seed; identical on every rank
    model = network.Classify_node(margs).to(device)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    trainer = train.GDTrainer(model, batch, lr=0.01, weight_decay=5e-4, dedup=not args.no_dedup, prune_unused_rows=args.prune_unused_rows)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step()
    barrier()
    events = []  # HIP-event pairs around the SpMM launches of every 4th step of the timed region (each marker costs ~1 us)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.PROFILE = events if i % 4 == 0 else None
        loss = trainer.step()
    ops.PROFILE = None
    barrier()
    dt = time.perf_counter() - t0
    # a few more steps, outside the timed region, with events around the hand-written GEMM launches (secondary figures:
    # their markers would cost the headline 1 %)
    ops.PROFILE_GEMM = []
    for _ in range(min(args.steps, 5)):
        trainer.step()
    torch.cuda.synchronize()
    gemm_events, ops.PROFILE_GEMM = ops.PROFILE_GEMM, None
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    edges_per_step = 4.0 * batch.nnz
    if trainer.sub is not None:   # two full SpMMs (layer 0) + the own-node rows of A_hat twice (layer 1 forward / backward)
        edges_per_step = 2.0 * batch.nnz + 2.0 * int(trainer.sub.f.col.numel())
    edges = torch.tensor([edges_per_step * args.steps], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(edges)
    dt, total_edges = float(tmax.item()), float(edges.item())

    # SpMM roofline: algorithmic bytes of one launch / mean HIP-event duration of the launches in the timed region
    H, R = args.hidden, batch.n_rows
    bytes_spmm = 4 * H * R + 4 * H * R + 8 * batch.nnz + 4 * (R + 1)
    # (the launches of the LDS-window kernel only: layer 0's forward on the de-duplicated table runs the direct-gather
    # variant, whose operand is a 40-MB table, not an [R x H] matrix)
    durs_ms = [a.elapsed_time(b) for a, b, kind in events if kind == "tile"]
    spmm_ms = float(np.mean(durs_ms)) if durs_ms else float("nan")
    achieved = bytes_spmm / (spmm_ms * 1e-3) / 1e9
    # HBM traffic of one SpMM launch from the committed PMC profile of this same command (rocprofv3 --pmc, separate
    # passes; gfx950 correction: FETCH_SIZE counts half of a wide coalesced read; both counters in KiB)
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_spmm_tile_kernel.json")
    if args.workload == "S-pubmed" and H == 512 and os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pmc = json.load(f)
        traffic = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
    # the hand-written MFMA GEMM kernels of the step (secondary: the step's dominant kernel class by time, not by launch):
    # bf16 flops actually issued (three products per fp32 product) / mean HIP-event duration, against the dense bf16 peak
    by_kernel = {}
    for a_ev, b_ev, name, flops in gemm_events:
        by_kernel.setdefault(name, []).append((a_ev.elapsed_time(b_ev), flops))
    gemm_summary = {name: {"launches_timed": len(v), "avg_launch_us": float(np.mean([d for d, _ in v])) * 1e3,
                           "bound": "mfma", "unit": "TFLOP/s (bf16, 3 products per fp32 product)",
                           "achieved": float(np.sum([f for _, f in v]) / (np.sum([d for d, _ in v]) * 1e-3) / 1e12),
                           "peak": 2500.0,
                           "frac": float(np.sum([f for _, f in v]) / (np.sum([d for d, _ in v]) * 1e-3) / 1e12 / 2500.0)}
                    for name, v in by_kernel.items()}
    # device-to-device copy ceiling of this GPU, same process, after the timed region (read + write bytes / time)
    src = torch.empty(256 << 20, dtype=torch.float32, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(10):
        dst.copy_(src)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del src, dst
    out = {
        "metric": "edges aggregated/sec (GCN fwd+bwd) on coarsened subgraphs",
        "value": total_edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: variation_neighborhoods r={WORKLOADS[args.workload][4]}, extra-node subgraphs, one block-diagonal "
                               f"union per GPU, 2-layer GCN hidden {H}, GD step + Adam", "parallelism": f"dp{world}",
                   "dense_gemm": ("fp32 operands split to 3 bf16 products on the MFMA pipe, fp32 accumulate (rel err ~5e-6 vs fp64): "
                                  "hand-written kernels gemm_nt.hip (X@W^T, dH@W with the previous layer's epilogue backward fused) and "
                                  "gemm_atb.hip (dH^T@X, split-K with a fixed-order sum); library GEMM only where K % 32 != 0 or the "
                                  "output is a few columns wide") if args.gemm_precision == "high" else "hipBLASLt fp32 MFMA",
                   "layer0_features": f"de-duplicated table ({info['nodes']} rows) + row indirection in the SpMM" if trainer.dedup
                   else "materialised union rows",
                   **info},
        "roofline": {"kernel": "spmm_tile_kernel<VEC=4,B=4,MPR=16> (CSR SpMM, LDS row windows, H=%d, f32)" % H, "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_spmm, "avg_launch_us": spmm_ms * 1e3,
                     "launches_timed": len(durs_ms), "spmm_edges_per_s": batch.nnz / (spmm_ms * 1e-3),
                     "copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs},
        "gemm_kernels": gemm_summary,
        "loss": float(loss),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch, sd0, 2)
        # the contraction step of the same graph on the host: the C restatement (one core), next to the HIP time above
        from oracle import coarsen_oracle as corc
        Wc, Ukc, lkc, rc = coarsen_inputs
        t0 = time.time()
        corc.coarsen_oracle(Wc, K=10, r=rc, Uk=Ukc.copy(), lk=lkc.copy())
        out["cpu_baseline"]["coarsen_port_s"] = round(time.time() - t0, 3)
        out["cpu_baseline"]["coarsen_port_cores"] = 1
        out["cpu_baseline"]["coarsen_reference_probe"] = "profiles/reference_probe_timings.json (reference Python, 8 cores: 17.97 s at this size)"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
