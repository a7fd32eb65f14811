"""CPU tier: properties of the BUILT gfx950 code objects that the kernels' correctness or speed rests on, read from the library's
metadata and disassembly (no GPU): a compiler update that breaks one of them fails the build check instead of a run on the GPU.

* the hand-counted `s_waitcnt vmcnt(N)` loops of gemm_atb.hip / gemm_nt.hip hold inline-asm loads in flight: safe only while hipcc
  neither spills nor copies those registers -- no VGPR spill, no scratch in these kernels;
* the default instantiations of the whole-subgraph SpMM kernel run at their designed occupancy: no spill, no scratch;
* the greedy selection's publish step relies on global_* accesses retiring in order on vmcnt: no flat_ load or atomic in that
  kernel, flat_ stores only for its write-through result words (coarsen.hip publish_top).
"""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fit-gnn_amd", "lib", "libfitgnn_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("ROCm's llvm tools are not installed")
    d = tmp_path_factory.mktemp("co")
    shutil.copy(LIB, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL)
    files = sorted(glob.glob(str(d / "lib.so.*gfx950*")))
    assert files, "no gfx950 code object in libfitgnn_hip.so"
    meta = {}
    for f in files:
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], check=True, stdout=subprocess.PIPE, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:|\n\s*- \.args:", txt):
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else None  # noqa: E731
            meta[name.group(1)] = dict(file=f, vgpr=get("vgpr_count"), vgpr_spill=get("vgpr_spill_count"), sgpr_spill=get("sgpr_spill_count"),
                                       scratch=get("private_segment_fixed_size"))
    return meta


def _kernels(meta, pattern):
    hits = {k: v for k, v in meta.items() if re.search(pattern, k)}
    assert hits, f"no kernel matches {pattern!r}: renamed?"
    return hits


def test_counted_wait_gemm_kernels_neither_spill_nor_use_scratch(code_objects):
    for pat in (r"gemm_atb_kernel", r"gemm_nt_kernel"):
        for name, m in _kernels(code_objects, pat).items():
            assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)


def test_exact_gemm_kernels_do_not_spill(code_objects):
    hits = _kernels(code_objects, r"gemm_f32_kernel")
    assert len(hits) >= 12   # five tile shapes x the three operand forms (64 x 512 for the split-k form only)
    for name, m in hits.items():
        assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)


def test_default_spmm_instantiations_keep_their_occupancy(code_objects):
    """spmm_block_kernel<XROW, EPI_BWD, NOEPI, TWO>: the three forms of the timed S-products step -- layer 0 on the table
    <true,false,false,false> (72 VGPR: 7 workgroups per CU), the plain product <false,false,true,false>, the two-hop backward
    <true,false,true,true> (<= 80 VGPR: 6 per CU) -- and the row-streaming kernels."""
    want = {r"spmm_block_kernelILb1ELb0ELb0ELb0E": 72, r"spmm_block_kernelILb0ELb0ELb1ELb0E": 72, r"spmm_block_kernelILb1ELb0ELb1ELb1E": 80,
            r"spmm_rows_compact_kernel": 64, r"two_hop_rows_kernel": 64}
    for pat, max_vgpr in want.items():
        for name, m in _kernels(code_objects, pat).items():
            assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)
            assert m["vgpr"] <= max_vgpr, (name, m)


def test_greedy_selection_uses_no_flat_access(code_objects):
    (name, m), = _kernels(code_objects, r"greedy_select_kernel").items()
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, m["file"]], check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    assert "global_load" in dis or "global_store" in dis, "disassembly of the kernel not found"
    # the set's members, lists and queue slots go through global_* accesses; the only flat_ instructions are the handful of
    # write-through (sc0 sc1) stores of the kernel's result words at its start and end
    assert not re.search(r"\bflat_(load|atomic)", dis), "publish_top's ordering argument needs global_* accesses"
    flat = re.findall(r"\bflat_store\S*[^\n]*", dis)
    assert len(flat) <= 8 and all("sc0 sc1" in f for f in flat), flat
    assert m["scratch"] == 0, m


def test_graph_step_kernels_keep_their_accumulators_in_registers(code_objects):
    """Round 4's graph-level step: narrow_atb_kernel<KT> holds 4 x KT + 4 running sums and eight rows' loads per thread, dense_narrow_k_kernel
    four rows' dot products, the batch assembly and the accumulating Adam are pure streams -- none may spill or use scratch."""
    for pat in (r"narrow_atb_kernel", r"dense_narrow_k_kernel", r"pool_head_kernel", r"pool_head_bwd_kernel", r"adam_flat_acc_kernel",
                r"batch_offsets_kernel", r"batch_gather_kernel", r"appnp_units_kernel", r"appnp_blocks_kernel", r"appnp_lds_kernel"):
        for name, m in _kernels(code_objects, pat).items():
            assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)
