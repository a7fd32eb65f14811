"""GPU tier: the RCCL ("nccl") branch of the data-parallel step, executed for real on the one GPU a test box has.

A one-rank RCCL group runs the very calls an N-rank job makes -- init_process_group("nccl", device_id=...), the parameter
broadcast at construction, the asynchronous all-reduce of the late gradient bucket launched from an autograd hook, the first
layer's bucket after the backward pass, the loss riding in the buffer's tail -- and a sum over one rank is the identity, so the
step must equal the non-distributed step BIT FOR BIT (run.py:184-204 is what makes the shard legal).  The N > 1
arithmetic is covered by the world-2 gloo tests (tests/test_dp_gloo.py, tests/test_gpu_callers.py); this file is what makes the
"nccl" code path something that has run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _star_batch(seed=0, n=3000, e=9000, clusters=300, F=40, C=5):
    from fitgnn_amd import data as fdata

    ei = torch.from_numpy(fdata.synthetic_graph(n, e, seed=seed)).cuda()
    rng = np.random.default_rng(seed)
    assign = rng.integers(0, clusters, size=n)
    assign[:clusters] = np.arange(clusters)
    sub = fdata.assemble_subgraphs_torch(ei, n, torch.from_numpy(assign).cuda(), clusters, extra_node=True, layout="star")
    X = rng.random((n, F), dtype=np.float32)
    y = rng.integers(0, C, size=n)
    return fdata.SubgraphBatch(sub, X, y, np.ones(n, dtype=bool), device="cuda")


@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_step_over_a_one_rank_rccl_group_equals_the_plain_step_bit_for_bit(dropout):
    from fitgnn_amd import network, train

    assert torch.cuda.is_available()
    dist = torch.distributed
    assert not dist.is_initialized(), "another test left a process group behind"
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        batch = _star_batch()
        args = argparse.Namespace(num_layers1=2, layer_name="GCNConv", num_features=40, hidden=64, num_classes=5, dropout=dropout)
        torch.manual_seed(3)
        m_plain = network.Classify_node(args).cuda()
        m_dist = network.Classify_node(args).cuda()
        m_dist.load_state_dict(m_plain.state_dict())
        # (capture=False: a captured step draws its dropout seeds from a device-resident bank, the distributed step from torch's generator)
        t_plain = train.GDTrainer(m_plain, batch, lr=0.01, weight_decay=5e-4, capture=False)
        assert not t_plain.dist                     # a one-rank default group alone does not switch the collectives on
        t_dist = train.GDTrainer(m_dist, batch, lr=0.01, weight_decay=5e-4, process_group=dist.group.WORLD)
        assert t_dist.dist and t_dist._split > 0    # two buckets: the late one leaves from the autograd hook
        t_dist.comm_events = []
        for step in range(4):
            torch.manual_seed(100 + step)           # the dropout seeds of a step are drawn from torch's generator
            a = t_plain.step()
            torch.manual_seed(100 + step)
            b = t_dist.step()
            assert t_dist._work is None, "the late bucket's work handle was not waited for"
            assert torch.equal(a, b), (step, float(a), float(b))
        torch.cuda.synchronize()
        for (k, v), (_, w) in zip(m_plain.state_dict().items(), m_dist.state_dict().items()):
            assert torch.equal(v, w), k
        assert torch.equal(t_plain.opt.m, t_dist.opt.m) and torch.equal(t_plain.opt.v, t_dist.opt.v)
        assert len(t_dist.comm_events) == 4 and all(x.elapsed_time(y) >= 0 for x, y in t_dist.comm_events)
        # the loss slot of the flat buffer went through the all-reduce with the late bucket
        assert float(t_dist.flat.tail[0]) == float(b)
    finally:
        dist.destroy_process_group()


def test_bench_shard_mode_steps_one_rank_of_a_job_and_its_shares_add_up(tmp_path):
    """`bench.py --shard K/N` (the single-GPU scaling evidence: tools/shard_curve.py): rank K's shard is built as --gpus N builds it,
    stepped alone over a one-rank RCCL group with the job's train count in the loss scale.  The ranks' first-step loss shares add up
    to the unsharded run's loss (dropout off), the shards' nnz' to the union's, and the line carries the all-reduce time."""
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--workload", "S-pubmed", "--steps", "1", "--warmup", "0", "--dropout", "0", "--no-cpu-baseline", "--no-bf16x3", "--no-all-rows",
              "--no-pruned"]
    lines = {}
    for tag, extra in (("all", []), ("0/2", ["--shard", "0/2"]), ("1/2", ["--shard", "1/2"])):
        res = subprocess.run([sys.executable, bench] + common + extra, env=env, cwd=tmp_path, check=True, timeout=900,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        lines[tag] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    whole, a, b = lines["all"], lines["0/2"], lines["1/2"]
    for k, line in ((0, a), (1, b)):
        emu = line["config"]["emulated"]
        assert emu["rank"] == k and emu["ranks"] == 2 and line["config"]["backend"] == "nccl"
        assert line["allreduce_ms"] is not None and line["allreduce_ms"] >= 0 and line["allreduce_bytes"] > 0
        assert line["config"]["partition_fingerprint"] == whole["config"]["partition_fingerprint"]
    assert a["config"]["owner_fingerprint"] == b["config"]["owner_fingerprint"]
    assert a["config"]["nnz_prime"] + b["config"]["nnz_prime"] == whole["config"]["nnz_prime"]
    assert abs(a["loss"] + b["loss"] - whole["loss"]) <= 1e-5 * abs(whole["loss"]), (a["loss"], b["loss"], whole["loss"])
    # heaviest/N names the rank with the largest weight
    w = a["config"]["emulated"]["rank_weights"]
    assert max(w) <= 1.02 * min(w)


@pytest.mark.parametrize("layer", ["GATConv", "APPNP"])
def test_bench_lines_of_the_other_operators(tmp_path, layer):
    """`bench.py --layer GATConv | APPNP` (north_star names GCN / GAT / APPNP): the same line shape as the headline -- the metric, a
    step-weighted roofline over the operator's own kernels with per-kind entries, the CPU oracle of the same step beside it."""
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, bench, "--layer", layer, "--workload", "S-cora", "--steps", "8", "--warmup", "2"], env=env, cwd=tmp_path,
                         check=True, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["metric"].startswith("edges aggregated/sec") and line["value"] > 0 and line["config"]["layer"] == layer
    kinds = {l["kind"] for l in line["roofline"]["launches"]}
    if layer == "GATConv":
        assert {"gat_scores", "gat_edge_softmax", "gat_aggregate", "gat_sddmm", "gat_softmax_bwd", "gat_aggregate_t"} <= kinds
    else:
        # (the K steps of the subgraphs that fit a wavefront's LDS: one launch each way; a per-step launch only where larger ones exist)
        assert {"appnp_units", "appnp_units_t"} <= kinds <= {"appnp_units", "appnp_units_t", "appnp_step", "appnp_step_t"}
        per = {l["kind"]: l["launches_per_step"] for l in line["roofline"]["launches"]}
        assert abs(per["appnp_units"] - 1) < 1e-9 and all(abs(v - 10) < 1e-9 for k, v in per.items() if k.startswith("appnp_step"))
    assert 0 < line["roofline"]["frac"] < 1.5 and line["roofline"]["bound"] == "hbm"
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert np.isfinite(line["loss"])
