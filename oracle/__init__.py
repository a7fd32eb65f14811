"""CPU oracle (test infrastructure only).  See oracle/c/fitgnn_oracle.c and oracle/*.py headers.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
