"""Helpers to read tests/golden/coarsen_<graph>.npz (layout: tests/golden/manifest.json)."""
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def graph_names():
    return sorted({c["name"] for c in manifest()["cases"]})


def cases():
    return [(c["name"], c["r"]) for c in manifest()["cases"] if "error" not in c]


class Golden:
    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN, f"coarsen_{name}.npz"))
        d = self.d
        self.N = len(d["W_indptr"]) - 1
        self.W = sp.csr_matrix((d["W_data"], d["W_indices"], d["W_indptr"]), shape=(self.N, self.N))
        self.Uk = d["Uk"] if d["Uk"].size else None
        self.lk = d["lk"] if d["lk"].size else None
        self.X = d["X"]
        self.K = int(d["K"])

    @staticmethod
    def rp(r):
        return f"r{int(round(r * 100)):02d}_"

    def A0(self):
        """Level-1 spectral matrix exactly as coarsening_utils.py:76-81 builds it from (Uk, lk)."""
        if self.Uk is None:
            return self.d["L0_A"]
        lk = self.lk.copy()
        mask = lk < 1e-10
        lk[mask] = 1
        lsinv = lk ** (-0.5)
        lsinv[mask] = 0
        return self.Uk[:, : self.K] @ np.diag(lsinv[: self.K])

    def n_levels(self, r):
        return int(self.d[self.rp(r) + "n_levels_recorded"])

    def level(self, r, li):
        """Inputs + recorded outputs of contraction level li (0-based) for ratio r."""
        d, p = self.d, self.rp(r) + f"L{li}_"
        if li == 0:
            W, A, cost0 = self.W, self.A0(), d["L0_cost0"]
            dw = np.ravel(W.sum(axis=0))
        else:
            n = len(d[p + "W_indptr"]) - 1
            W = sp.csr_matrix((d[p + "W_data"], d[p + "W_indices"], d[p + "W_indptr"]), shape=(n, n))
            A, cost0, dw = d[p + "A"], d[p + "cost0"], d[p + "dw"]
        return dict(W=W, A=A, dw=dw, cost0=cost0, r_cur=float(d[p + "r_cur"]),
                    trace_cand=d[p + "trace_cand"], trace_cost=d[p + "trace_cost"],
                    trace_off=d[p + "trace_off"], trace_mem=d[p + "trace_mem"],
                    sel_off=d[p + "sel_off"], sel_mem=d[p + "sel_mem"],
                    iC=sp.csc_matrix((d[p + "iC_data"], d[p + "iC_indices"], d[p + "iC_indptr"]),
                                     shape=tuple(d[p + "iC_shape"])))

    def final(self, r):
        d, p = self.d, self.rp(r)
        shape = tuple(d[p + "C_shape"])
        n = int(d[p + "Gc_N"])
        return dict(C=sp.csc_matrix((d[p + "C_data"], d[p + "C_indices"], d[p + "C_indptr"]), shape=shape),
                    assign=d[p + "assign"], CX64=d[p + "CX64"],
                    GcW=sp.csr_matrix((d[p + "GcW_data"], d[p + "GcW_indices"], d[p + "GcW_indptr"]), shape=(n, n)),
                    n_mapping_dicts=int(d[p + "n_mapping_dicts"]))
