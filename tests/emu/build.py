"""Builds tests/emu/libwt_emu.so (CPU emulator of the HIP kernels, test-only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    so = os.path.join(HERE, "libwt_emu.so")
    srcs = [os.path.join(HERE, "wt_emu.cpp"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_core.h"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_plan.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                           "-Wno-unused-function", "-o", so, srcs[0], "-lm"])
    return so


if __name__ == "__main__":
    print(build(force=True))
