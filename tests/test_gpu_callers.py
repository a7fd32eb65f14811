"""GPU tier: the callers either side of the hot path against literal restatements of the reference's Python
(oracle/gs_oracle.py): Gc label / mask / edge assembly (utils.py:705-775), device subgraph assembly
(f1: utils.py:186-267, :683-703), and the data-parallel command line (e: run.py:177-215 under torch.distributed)."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import Golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cora_union(r, names=("cora_giant", "cora26", "cora9", "cora2"), seed=0):
    """The golden Cora components as ONE dataset (block-diagonal, components in size order as utils.py:146 sorts them),
    with the reference's recorded C and Gc.W per component."""
    from fitgnn_amd import coarsening, pipeline

    rng = np.random.default_rng(seed)
    gs = [Golden(n) for n in names]
    off, comps, xs = 0, [], []
    co = pipeline.Coarsened()
    co.components, co.C_list, co.Gc_list = [], [], []
    ref_C, ref_GcW = [], []
    for g in gs:
        H = coarsening.Graph(g.W)
        H.info = {"orig_idx": list(range(off, off + g.N))}
        co.components.append(H)
        comps.append((H.info["orig_idx"], g.W))
        if g.N > 10:   # utils.py:164-166
            fin = g.final(r)
            co.C_list.append(coarsening.CoarseningMatrix(fin["C"]))
            co.Gc_list.append(coarsening.Graph(fin["GcW"]))
            ref_C.append(fin["C"]); ref_GcW.append(fin["GcW"])
        xs.append(g.X.astype(np.float32))
        off += g.N
    F = min(x.shape[1] for x in xs)
    X = np.concatenate([x[:, :F] for x in xs])
    y = rng.integers(0, 7, size=off)
    return co, comps, ref_C, ref_GcW, X, y, [g.N for g in gs]


@pytest.mark.parametrize("r", [0.3, 0.5, 0.7])
@pytest.mark.parametrize("case", ["mixed", "dense_labels", "val_only_small"])
def test_gc_assembly_matches_the_reference_loop(r, case):
    """pipeline.build_gc == utils.py:705-775 restated (oracle/gs_oracle.load_gc) on the golden Cora components with the
    reference's own C / Gc.W: pooled features bit for bit, argmax labels, the single-class mask rule, edges with the
    running offset in the reference's order, small components passed through or dropped."""
    from fitgnn_amd import pipeline
    from oracle import gs_oracle

    co, comps, ref_C, ref_GcW, X, y, sizes = _cora_union(r)
    N = sum(sizes)
    rng = np.random.default_rng(7)
    tm, vm = np.zeros(N, dtype=bool), np.zeros(N, dtype=bool)
    if case == "mixed":           # sparse labels: most clusters hold 0 or 1 labelled node
        tm[rng.choice(sizes[0], 140, replace=False)] = True
        vm[rng.choice(sizes[0], 300, replace=False)] = True
        tm[sizes[0] + 3] = True                                   # cora26: one train node
        vm[sizes[0] + sizes[1] + 2] = True                        # cora9: passes through un-coarsened (:754-769)
    elif case == "dense_labels":  # every node labelled: many clusters mix classes -> masked out (:727-730)
        tm[:] = True
        vm[rng.random(N) < 0.5] = True
    else:                         # a small component holding only val nodes; cora26 holds nothing and is dropped
        tm[rng.choice(sizes[0], 50, replace=False)] = True
        vm[sizes[0] + sizes[1]: sizes[0] + sizes[1] + sizes[2]] = True
        vm[N - 1] = True                                          # cora2 too
    data = pipeline.NodeData(torch.from_numpy(X), None, torch.from_numpy(y), torch.from_numpy(tm), torch.from_numpy(vm))
    args = argparse.Namespace(num_classes=7, normalize_features=False)
    gc = pipeline.build_gc(args, data, co, device="cuda")
    ref = gs_oracle.load_gc(comps, ref_C, ref_GcW, X, y, tm, vm, 7)
    assert np.array_equal(gc.x.cpu().numpy(), ref["features"])
    assert np.array_equal(gc.train_labels.cpu().numpy(), ref["train_labels"])
    assert np.array_equal(gc.val_labels.cpu().numpy(), ref["val_labels"])
    assert np.array_equal(gc.train_idx.cpu().numpy(), np.nonzero(ref["train_mask"])[0])
    assert np.array_equal(gc.val_idx.cpu().numpy(), np.nonzero(ref["val_mask"])[0])
    assert np.array_equal(gc.edge_index.cpu().numpy(), ref["edge"])
    if case == "dense_labels":
        assert ref["train_mask"].sum() < len(ref["train_mask"]), "the case must exercise the mixed-class rule"


def test_gc_assembly_refuses_a_graph_that_needs_no_coarsening():
    """utils.py:763: the first component with labelled nodes must be a coarsened one."""
    from fitgnn_amd import coarsening, pipeline

    g = Golden("cora9")
    H = coarsening.Graph(g.W)
    H.info = {"orig_idx": list(range(g.N))}
    co = pipeline.Coarsened()
    co.components, co.C_list, co.Gc_list = [H], [], []
    tm = np.zeros(g.N, dtype=bool); tm[0] = True
    data = pipeline.NodeData(torch.from_numpy(g.X.astype(np.float32)), None, torch.zeros(g.N, dtype=torch.long),
                             torch.from_numpy(tm), torch.zeros(g.N, dtype=torch.bool))
    with pytest.raises(Exception, match="does not need coarsening"):
        pipeline.build_gc(argparse.Namespace(num_classes=7, normalize_features=False), data, co, device="cuda")


@pytest.mark.parametrize("extra", [True, False])
@pytest.mark.parametrize("source", ["cora_giant", "random"])
def test_device_subgraph_assembly_matches_the_reference_loop(source, extra):
    """data.assemble_subgraphs_torch on the MI355X (+ SubgraphBatch's masks) == the reference's per-cluster construction
    (oracle/gs_oracle.cluster_subgraphs, utils.py:186-267) and its mask stamping (utils.py:683-703): node lists, own / extra
    flags, relabelled induced edges, train masks -- on the real Cora giant component with the reference's recorded
    partition, and on a random graph with a random partition (clusters of 1..n nodes, isolated clusters)."""
    from fitgnn_amd import data as fdata
    from oracle import gs_oracle

    if source == "cora_giant":
        g = Golden("cora_giant")
        coo = g.W.tocoo()
        ei = np.stack([coo.row, coo.col]).astype(np.int64)
        assign = g.final(0.5)["assign"].astype(np.int64)
        N = g.N
    else:
        N = 600
        ei = fdata.synthetic_graph(N, 1500, seed=5)
        rng = np.random.default_rng(3)
        assign = rng.integers(0, 170, size=N)
        assign[:170] = np.arange(170)
    n = int(assign.max()) + 1
    sub = fdata.assemble_subgraphs_torch(torch.from_numpy(ei).cuda(), N, assign, n, extra_node=extra, chunk_rows=1000)
    ref = gs_oracle.cluster_subgraphs(ei, N, assign, extra)
    ptr = sub["ptr"].cpu().numpy()
    node_id, core, e = sub["node_id"].cpu().numpy(), sub["core"].cpu().numpy(), sub["edge_index"].cpu().numpy()
    assert len(ref) == n and int(ptr[-1]) == sum(len(s["orig_idx"]) for s in ref)
    assert len(set(zip(e[0].tolist(), e[1].tolist()))) == e.shape[1], "no duplicate edges"
    owner = np.searchsorted(ptr, e[0], side="right") - 1
    rng = np.random.default_rng(1)
    tm, vm, te = rng.random(N) < 0.3, rng.random(N) < 0.3, rng.random(N) < 0.3
    batch = fdata.SubgraphBatch(sub, np.zeros((N, 4), dtype=np.float32), np.zeros(N, dtype=np.int64), tm, device="cuda")
    btm = batch.train_mask.cpu().numpy()
    for c, s in enumerate(ref):
        r0, r1 = int(ptr[c]), int(ptr[c + 1])
        assert np.array_equal(node_id[r0:r1], s["orig_idx"])
        assert np.array_equal(core[r0:r1], ~np.isin(s["orig_idx"], s["actual_ext"]))
        mine = e[:, owner == c] - r0
        assert set(zip(mine[0].tolist(), mine[1].tolist())) == set(zip(s["edge_index"][0].tolist(), s["edge_index"][1].tolist()))
        assert mine.shape[1] == s["edge_index"].shape[1]
        tr, _, _ = gs_oracle.subgraph_split_masks(s, tm, vm, te, extra)
        assert np.array_equal(btm[r0:r1], tr)


def test_cli_data_parallel_run_equals_the_single_process_run(tmp_path):
    """BASELINE.json config 4 through the kept command line: `main.py ... --gradient_method GD` launched as two ranks
    (gloo over one GPU: a rehearsal of the RCCL path, same code) trains to the weights of the one-process run
    (dropout off so that the two are comparable: the union's rows sit at different offsets in the shards)."""
    main_py = os.path.join(ROOT, "fit-gnn_amd", "main.py")
    common = ["--dataset", "synthetic-cora", "--hidden", "64", "--seed", "0", "--runs", "1", "--epochs2", "6", "--train_fitgnn",
              "--exp_setup", "Gs_train_2_Gs_infer", "--extra_node", "--normalize_features", "--dropout", "0.0",
              "--gradient_method", "GD"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = tmp_path / "one"; one.mkdir()
    subprocess.run([sys.executable, main_py] + common + ["--output_dir", "o"], cwd=one, env=env, check=True, timeout=600,
                   stdout=subprocess.DEVNULL)
    two = tmp_path / "two"; two.mkdir()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, main_py] + common + ["--output_dir", "o"], cwd=two, stdout=subprocess.DEVNULL,
                              env=dict(env, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port), FITGNN_DIST_BACKEND="gloo")) for k in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a = torch.load(one / "save" / "node_cls" / "o" / "model.pt", map_location="cpu")
    b = torch.load(two / "save" / "node_cls" / "o" / "model.pt", map_location="cpu")
    assert a.keys() == b.keys()
    # Adam divides by sqrt(v): an entry whose gradient is at rounding-noise level moves by up to lr per step in either run, so
    # the comparison is "almost every entry to 1e-5, no entry beyond a few noise-driven steps"
    for k in a:
        d, top = (a[k] - b[k]).abs(), float(a[k].abs().max())
        assert float(d.max()) <= 2e-3 * top + 1e-6, k
        assert float((d > 1e-5 * top + 1e-7).float().mean()) < 0.02, k
    ra = (one / "results" / "synthetic-cora.csv").read_text().splitlines()
    rb = (two / "results" / "synthetic-cora.csv").read_text().splitlines()
    assert len(ra) == len(rb) == 2, "rank 0 alone writes the results row"


def test_bench_two_ranks_report_the_single_rank_loss(tmp_path):
    """bench.py's data-parallel path end to end, as the driver launches it: `python bench.py --gpus 2` spawns its two ranks itself
    (a parent that never touches the GPU; gloo over this one GPU = a rehearsal of the RCCL path, same code): rank 0 coarsens and
    broadcasts the partition, every rank shards it BEFORE assembling and builds its own clusters only, steps on its shard, one
    gradient all-reduce per step -- the global loss of the step equals the one-rank run's (dropout off: its hash is keyed on a
    rank's own row numbers; ONE step: Adam's first update is g / |g| per weight, so gradients at rounding-noise level move their
    weights by +-lr in either run and later losses agree only to ~1e-3), and the line carries the ranks' nnz' and the all-reduce
    time."""
    import json

    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FITGNN_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--workload", "S-pubmed", "--steps", "1", "--warmup", "0", "--dropout", "0", "--no-cpu-baseline", "--no-bf16x3", "--no-all-rows"]
    lines = {}
    for n in (1, 2):
        res = subprocess.run([sys.executable, bench, "--gpus", str(n)] + common, env=env, cwd=tmp_path, check=True, timeout=900,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        lines[n] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "dp2"
    assert one["config"]["partition_fingerprint"] == two["config"]["partition_fingerprint"], "the two runs coarsened to different partitions"
    assert abs(two["loss"] - one["loss"]) <= 1e-5 * abs(one["loss"]), (one["loss"], two["loss"])
    shards = two["config"]["shard_nnz_prime"]
    assert len(shards) == 2 and sum(shards) == one["config"]["nnz_prime"] == two["config"]["nnz_prime"]
    assert max(shards) <= 1.05 * min(shards)
    assert two["allreduce_ms"] is not None and two["allreduce_ms"] > 0
    assert len(one["roofline"]["launches"]) == 4 and [l["launch"] for l in one["roofline"]["launches"]] == [
        "layer0_forward", "layer1_forward", "layer1_backward", "layer0_backward"]
