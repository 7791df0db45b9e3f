"""MeanReduction's division on the device (csrc/wt_delta.h wt_div_n: s / n for the track count n as one multiply and two fused
multiply-adds) against the division the reference computes (reducers.c:375-401: sum / count), bit for bit: every count the
difference-array launches accept (1 .. 32767), quotients on and beside the doubles' rounding boundaries, both signs.  Compiled for
the host from the same source (tests/emu)."""
import ctypes as C

import pytest

from emu import build as emu_build


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(emu_build.build())
    L.wtemu_div_n_mismatches.restype = C.c_longlong
    L.wtemu_div_n_mismatches.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong]
    return L


def test_every_count_a_few_hundred_quotients(lib):
    assert lib.wtemu_div_n_mismatches(1, 32767, 200, 1) == 0       # 6.5e6 quotients x 5 neighbours x 2 signs


@pytest.mark.parametrize("n", [1, 2, 3, 7, 10, 100, 127, 128, 500, 1000, 4095, 4097, 32767])
def test_counts_of_the_configurations_in_depth(lib, n):
    assert lib.wtemu_div_n_mismatches(n, n, 400000, 7 + n) == 0
