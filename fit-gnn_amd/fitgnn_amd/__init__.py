"""fitgnn_amd -- MI355X-native implementation of FIT-GNN's coarsen-then-train hot path.

Host-side mirror of the reference's operator surfaces over libfitgnn_hip.so (C ABI: include/fitgnn_hip.h):
    fitgnn_amd.coarsening.coarsen      <- graph_coarsening/coarsening_utils.py:18  coarsen
    fitgnn_amd.nn.{GCNConv,...}        <- torch_geometric.nn layers resolved by network.py:13
    fitgnn_amd.network.Classify_node…  <- network.py:8-204
"""
from . import _lib  # noqa: F401

__all__ = ["coarsening", "nn", "network", "ops", "csr"]
