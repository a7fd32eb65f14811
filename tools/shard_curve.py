#!/usr/bin/env python3
"""The scaling evidence ONE MI355X can give: the heaviest rank's step of an N-rank data-parallel job, for
N = 1, 2, 4, 8, each timed alone on this GPU by `bench.py --shard heaviest/N` (a fresh child process per N: the shard is built exactly
as `--gpus N` builds it, the gradient all-reduce runs over RCCL in a group of one), and the model of DESIGN.md §5 fitted to it:

    step_ms(N) = c + s * nnz'_shard(N) / 1e6            (c: what does not shrink with N; s: ms per million edges of the shard)
    predicted job value(N) = 4 * nnz'_total / (step_ms(N) + allreduce_model_ms(N))

allreduce_model_ms(N): the measured cost of the two RCCL calls of a step in a one-rank group (launch path, `allreduce_ms`) plus a
ring term for N ranks over xGMI -- 2 (N - 1) hops of ~8 us latency + 2 (N - 1) / N * bytes / 100 GB/s achieved per link -- a MODEL, not a
measurement: no multi-GPU node was available.  usage (GPU box): python3 tools/shard_curve.py [out.json] [extra bench.py args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--shard", f"heaviest/{n}", "--no-cpu-baseline", "--no-bf16x3", "--no-all-rows",
           "--no-pruned"] + extra
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, check=True)
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    extra = sys.argv[2:] if out_path else sys.argv[1:]
    rows = []
    for n in (1, 2, 4, 8):
        line = run(n, extra)
        r = line["roofline"]
        rows.append({"ranks": n, "rank": line["config"]["emulated"]["rank"], "shard_nnz_prime": line["config"]["nnz_prime"],
                     "shard_union_rows": line["config"]["union_rows"], "share_of_weight": line["config"]["emulated"]["share_of_weight"],
                     "shard_ms": line["ms_per_step"], "spmm_ms": r.get("spmm_ms_per_step") or None, "gemm_ms": line["gemm_ms_per_step"],
                     "allreduce_ms_one_rank_group": line["allreduce_ms"], "allreduce_bytes": line["allreduce_bytes"],
                     "spmm_launches": [{k: l[k] for k in ("kind", "avg_us", "frac")} for l in r.get("launches", [])],
                     "loss_share": line["loss"], "steps": line["steps"]})
        print(json.dumps(rows[-1]), flush=True)
    total = rows[0]["shard_nnz_prime"]
    # least squares of step_ms on the shard's nnz' (four points)
    xs = [r["shard_nnz_prime"] / 1e6 for r in rows]
    ys = [r["shard_ms"] for r in rows]
    mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
    s = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    c = my - s * mx
    for r in rows:
        n = r["ranks"]
        ring = 0.0 if n == 1 else (2 * (n - 1) * 8e-3 + 2 * (n - 1) / n * r["allreduce_bytes"] / 100e9 * 1e3)
        r["allreduce_model_ms"] = (r["allreduce_ms_one_rank_group"] or 0.0) + ring
        r["predicted_job_value"] = 4.0 * total / ((r["shard_ms"] + r["allreduce_model_ms"]) * 1e-3)
        r["other_ms"] = r["shard_ms"] - (r["spmm_ms"] or 0.0) - r["gemm_ms"]
    base = rows[0]["predicted_job_value"]
    for r in rows:
        r["predicted_efficiency"] = r["predicted_job_value"] / (base * r["ranks"])
    out = {"what": "heaviest rank of an N-rank job stepped alone on one MI355X (bench.py --shard heaviest/N); PREDICTION of the N-GPU line, "
                   "not a measurement of it", "total_nnz_prime": total,
           "model": {"step_ms = c + s * nnz'_shard / 1e6": {"c_ms": c, "s_ms_per_million_edges": s}}, "rows": rows}
    txt = json.dumps(out, indent=1)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
