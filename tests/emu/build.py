"""Builds tests/emu/libwt_emu.so (CPU emulator of the HIP kernels, test-only)."""
import fcntl
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def _compile(so, deps, cmd_tail):
    """g++ to a temporary name, rename() into place, one build at a time (pytest-xdist workers share the checkout)."""
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in deps):
            return so                                   # another worker built it while this one waited
        tmp = "%s.tmp.%d" % (so, os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                               "-Wno-unused-function", "-Wno-unknown-pragmas", "-o", tmp] + cmd_tail)
        os.replace(tmp, so)
    return so


def build(force=False):
    so = os.path.join(HERE, "libwt_emu.so")
    srcs = [os.path.join(HERE, "wt_emu.cpp"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_core.h"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_plan.h"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_delta.h"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_walk.h"),
            os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc", "wt_mwalk.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    if force and os.path.exists(so):
        os.utime(srcs[0])
    return _compile(so, srcs, [srcs[0], "-lm"])


def build_variant(name, flags):
    """tests/emu/libwt_emu_<name>.so: the emulator compiled with extra flags (compile-time switches of the kernels' logic that are
    not the default, e.g. -DWT_DELTA_PARK=2), rebuilt when a source is newer."""
    so = os.path.join(HERE, "libwt_emu_%s.so" % name)
    csrc = os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc")
    srcs = [os.path.join(HERE, "wt_emu.cpp")] + [os.path.join(csrc, h) for h in ("wt_core.h", "wt_plan.h", "wt_delta.h", "wt_walk.h", "wt_mwalk.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    return _compile(so, srcs, list(flags) + [srcs[0], "-lm"])


def build_dropin(force=False):
    """tests/emu/libwt_dropin_emu.so: the product's drop-in layer (csrc/wt_iter_abi.cpp,
    csrc/wt_defaults.cpp) linked against the emulated pipeline (wt_pipe_emu.cpp + wt_emu.cpp)."""
    so = os.path.join(HERE, "libwt_dropin_emu.so")
    csrc = os.path.join(HERE, "..", "..", "wiggletools_amd", "csrc")
    srcs = [os.path.join(HERE, "wt_emu.cpp"), os.path.join(HERE, "wt_pipe_emu.cpp"),
            os.path.join(csrc, "wt_iter_abi.cpp"), os.path.join(csrc, "wt_defaults.cpp"), os.path.join(csrc, "wt_bigwig.cpp")]
    deps = srcs + [os.path.join(csrc, h) for h in ("wt_core.h", "wt_plan.h", "wt_delta.h", "wt_walk.h", "wt_mwalk.h", "wt_abi_common.h", "wt_abi_feeder.h", "wt_abi_reduce.h", "wt_abi_readers.h", "wt_abi_bwdev.h", "wt_abi_ops.h", "wt_abi_integrators.h", "wt_bufreader.h", "wt_inflate.h", "wt_bwdev_core.h",
                                                    "wt_mapop.h", "wt_bigwig_int.h")] + \
        [os.path.join(HERE, "..", "..", "include", "wiggletools_amd.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in deps):
        return so
    if force and os.path.exists(so):
        os.utime(srcs[0])
    return _compile(so, deps, srcs + ["-lm", "-lz", "-lpthread"])


if __name__ == "__main__":
    print(build(force=True))
    print(build_dropin(force=True))
