"""Registers and scratch memory of the built gfx950 kernels, read off the library's code objects (no GPU needed).

The emulator cannot see a spilled register, and a spill is not a correctness bug: in round 5 a run-time bound around the scans of
wt_delta_kernel took its Sum / Mean instantiation from 3 to 34 spilled registers -- every parity test stayed green, C2 lost 10 % and the
kernel moved 1.29 x instead of 1.036 x its algorithmic bytes through HBM (profiles/r05_delta_pass2_experiments.txt, H).  These
budgets are what the committed kernels use, with a little slack; a change that breaks one should know it."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels():
    import __graft_entry__ as g
    g.build()
    from wiggletools_amd import _lib
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm image not found")
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(_lib.LIB_PATH, so)
        subprocess.run([objdump, "--offloading", so], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for co in glob.glob(os.path.join(tmp, "lib.so.*gfx950*")):
            notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                def field(k, block=block):
                    m = re.search(r"\.%s:\s+(\d+)" % k, block)
                    return int(m.group(1)) if m else 0
                out[name.group(1)] = {"vgpr": field("vgpr_count"), "sgpr": field("sgpr_count"), "spill": field("vgpr_spill_count"),
                                      "scratch": field("private_segment_fixed_size"), "max_wg": field("max_flat_workgroup_size")}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.fixture(scope="module")
def kernels():
    k = _kernels()
    assert len(k) > 30, sorted(k)
    return k


def _delta(kernels, op, df, u=4):
    """wt_delta_kernel<op, df, u>: u = runs per lane and tile of the pass (round 6: Sum / Mean also exist with 2)."""
    name = "_Z15wt_delta_kernelILi%dELb%dELi%dEEv8WtParams" % (op, 1 if df else 0, u)
    assert name in kernels, [n for n in kernels if "delta" in n]
    return kernels[name]


def test_sum_mean_difference_array_kernels_fit_their_registers(kernels):
    """wt_delta_kernel<sum | mean>: 1024 lanes, so 128 registers per lane -- and next to nothing in scratch memory (it is written and read
    once per window by every lane, and that traffic reaches HBM)."""
    for op in (0, 2):
        for df in (False, True):
            for u in (4, 2):
                k = _delta(kernels, op, df, u)
                assert k["max_wg"] == 1024 and k["vgpr"] <= 128, k
                assert k["spill"] <= (8 if df else 6), (op, df, u, k)
                assert k["scratch"] <= 64, (op, df, u, k)


def test_squares_difference_array_kernels_spill_nothing(kernels):
    """wt_delta_kernel<var | stddev | cv>: 768 lanes = three wavefronts per SIMD = 168 registers per lane, none in scratch memory
    (at 1024 lanes / 128 registers the scans spilled 41: C3 57.5 against 55.5 ms)."""
    for op in (3, 4, 6):
        k = _delta(kernels, op, False)
        assert k["max_wg"] == 768 and k["vgpr"] <= 170, k
        assert k["spill"] == 0 and k["scratch"] == 0, (op, k)


def test_two_sample_difference_array_kernel_budget(kernels):
    """wt_delta_kernel<ttest> (round 6): 768 lanes, 168 registers; its scans (two sets' 128-bit sums per lane) and the Student tail
    spill -- a few hundred bytes of scratch written and read in the scan phases only: the passes over the runs touch none
    (read off the ISA: tools/kernel_asm.py 10 0).  The budget holds the regression class: a change that lets the spills grow
    past this or reach the pass shows up here."""
    for u in (4, 2):
        k = _delta(kernels, 10, False, u)
        assert k["max_wg"] == 768 and k["vgpr"] <= 170, k
        assert k["spill"] <= 120 and k["scratch"] <= 512, (u, k)


def test_walking_and_inflate_kernels_use_no_scratch(kernels):
    for name, k in kernels.items():
        if "wt_walk_kernel" in name or "wt_mwalk_kernel" in name or "wt_bw_inflate_kernel" in name:
            assert k["spill"] == 0 and k["scratch"] == 0, (name, k)
