"""Builds wiggletools_amd/csrc/libwiggletools_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libwiggletools_amd.so")
SRCS = ["wt_engine.hip", "wt_walk.hip", "wt_compress.hip", "wt_map.hip", "wt_synth.hip", "wt_bwdev.hip", "wt_defaults.cpp", "wt_iter_abi.cpp", "wt_bigwig.cpp", "wt_bwwrite.cpp"]
LIBS = ["-lz"]
DEPS = ["wt_abi_common.h", "wt_abi_feeder.h", "wt_abi_reduce.h", "wt_abi_readers.h", "wt_abi_bwdev.h", "wt_abi_ops.h", "wt_abi_integrators.h", "wt_core.h", "wt_delta.h", "wt_walk.h", "wt_mwalk.h", "wt_bufreader.h", "wt_bigwig_int.h", "wt_plan.h", "wt_devscope.h", "wt_pipe.h", "wt_mapop.h", "wt_inflate.h", "wt_bwdev_core.h", os.path.join("..", "..", "include", "wiggletools_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def build_variant(name, extra_flags):
    """Experiment helper: builds libwiggletools_amd_<name>.so with extra compiler flags
    (select it at run time with WTAMD_LIB=<path>)."""
    srcs = [os.path.join(HERE, s) for s in SRCS if os.path.exists(os.path.join(HERE, s))]
    out = os.path.join(HERE, "libwiggletools_amd_%s.so" % name)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + FLAGS + list(extra_flags) + srcs + LIBS + ["-o", out])
    return out


def build_engine_variant(name, extra_flags):
    """Experiment helper, faster than build_variant: recompiles only wt_engine.hip with extra flags and links it
    with the objects of the last build() (run build() first) into libwiggletools_amd_<name>.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, ".obj")
    cflags = [f for f in FLAGS if f != "-shared"]
    obj = os.path.join(objdir, "wt_engine_%s.o" % name)
    subprocess.check_call([hipcc] + cflags + list(extra_flags) + ["-c", os.path.join(HERE, "wt_engine.hip"), "-o", obj])
    others = [os.path.join(objdir, s + ".o") for s in SRCS if s != "wt_engine.hip"]
    out = os.path.join(HERE, "libwiggletools_amd_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + LIBS + ["-o", out])
    return out


def build_source_variant(src, name, extra_flags):
    """Experiment helper: recompiles ONE source with extra flags and links it with the objects of the last build() into
    libwiggletools_amd_<name>.so (e.g. build_source_variant("wt_bwdev.hip", "round8", ["-DWT_INF_ROUND=8"]))."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, ".obj")
    cflags = [f for f in FLAGS if f != "-shared"]
    obj = os.path.join(objdir, "%s_%s.o" % (src, name))
    subprocess.check_call([hipcc] + cflags + list(extra_flags) + ["-c", os.path.join(HERE, src), "-o", obj])
    others = [os.path.join(objdir, s + ".o") for s in SRCS if s != src]
    out = os.path.join(HERE, "libwiggletools_amd_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + LIBS + ["-o", out])
    return out


def build(force=False, verbose=False):
    """Every source to its own object (in parallel: wt_engine.hip alone takes minutes), then one link."""
    srcs = [os.path.join(HERE, s) for s in SRCS if os.path.exists(os.path.join(HERE, s))]
    deps = srcs + [os.path.join(HERE, d) for d in DEPS]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        if verbose:
            print("wiggletools_amd: libwiggletools_amd.so is newer than its %d sources / headers: REUSED (force=True recompiles)" % len(deps))
        return SO
    if verbose:
        print("wiggletools_amd: COMPILING %d sources for gfx950 with hipcc" % len(srcs))
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, ".obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]

    def up_to_date(obj):
        """The object is newer than its source and every header the compiler saw last time (-MMD dependency file)."""
        dep = obj + ".d"
        if force or not (os.path.exists(obj) and os.path.exists(dep)):
            return False
        try:
            words = open(dep).read().replace("\\\n", " ").split()
        except OSError:
            return False
        t = os.path.getmtime(obj)
        files = [w for w in words[1:] if not w.endswith(":")]
        return bool(files) and all(os.path.exists(f) and os.path.getmtime(f) <= t for f in files)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if up_to_date(obj):
            if verbose:
                print("wiggletools_amd: %s unchanged: object REUSED" % os.path.basename(src))
            return obj
        cmd = [hipcc] + cflags + ["-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(len(srcs)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + LIBS + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
