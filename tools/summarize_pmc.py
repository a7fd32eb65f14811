#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes written by tools/profile_round.sh: for every kernel of the train step
(name filter below) the mean counter value per dispatch over the dispatches of the timed steps, and for the SpMM the HBM
traffic per launch as MI355X_MICROARCH.md prescribes (gfx950: FETCH_SIZE counts half the bytes of a wide coalesced read;
both counters are in KiB): traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

Output (what bench.py's pmc_traffic reads): {"config": the configuration of the profiled run (bench.py's `run_config` of the stats
pass + the commit named in $FITGNN_COMMIT), "kinds": launch kind of ops.OpConfig.profile -> the kernels that make it up, "kernels":
per-kernel counters}.  A kind whose kernels were not seen in the passes fails the summary loudly instead of leaving a hole."""
import csv
import glob
import json
import os
import sys

KEEP = ("two_hop_rows_kernel", "spmm_tile_kernel", "spmm_block_kernel", "spmm_rows_compact_kernel", "spmm_stream_kernel", "spmm_gather_kernel", "gemm_f32_kernel", "gemm_nt_kernel",
        "gemm_atb_kernel", "epilogue_bwd_kernel", "segment_sum_kernel", "head_rows_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            i = name.index(k)
            j = name.find("(", i)
            return name[i:j if j > 0 else None].strip()
    return None


# launch kind (ops.py: cfg.profile entries) -> substrings naming the kernels of one launch of that kind on the whole-subgraph path
KINDS = {"table": ["spmm_block_kernel<true, false, false, false>"],
         "tile": ["spmm_block_kernel<false, false, true, false>"],
         "two_hop": ["spmm_block_kernel<true, false, true, true>", "two_hop_rows_kernel"]}


def main(root):
    out = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (k, c), v in acc.items():
                v = v[len(v) // 3:]   # skip the warm-up steps' dispatches
                out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "dispatches": len(v)}
    for k, c in out.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            h, m = c["TCC_HIT_sum"]["mean"], c["TCC_MISS_sum"]["mean"]
            c["l2_hit_rate"] = h / max(h + m, 1.0)
    cfg = {}
    try:
        with open(os.path.join(root, "bench_under_rocprof.json")) as fh:
            line = [l for l in fh if l.startswith("{")][-1]
        cfg = dict(json.loads(line).get("run_config", {}))
    except (OSError, IndexError, ValueError):
        pass
    cfg["commit"] = os.environ.get("FITGNN_COMMIT", "unknown")
    kinds = {}
    for kind, pats in KINDS.items():
        names = []
        for pat in pats:
            hit = [k for k in out if k.startswith(pat) and "hbm_bytes_per_launch" in out[k]]
            if len(hit) != 1:
                raise SystemExit(f"launch kind {kind!r}: kernel {pat!r} matched {hit} in the PMC passes -- update KINDS")
            names.append(hit[0])
        kinds[kind] = names
    print(json.dumps({"config": cfg, "kinds": kinds, "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
