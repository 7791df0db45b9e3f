#!/usr/bin/env python
"""ISA of ONE instantiation of wt_delta_kernel in seconds instead of the minutes wt_engine.hip takes: the kernel's text + the
headers in a scratch file, one explicit instantiation, hipcc -S.   tools/kernel_asm.py 10 0 [-DFLAG ...] > /tmp/k.s
(10 = WT_OP_TTEST, 2 = mean, 6 = stddev ...; second argument: DF)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "wiggletools_amd/csrc/wt_engine.hip")).read()
head = src[:src.index("// ---------------------------------------------------------------------------\n// kernels")]
a = src.index("#define WT_DELTA_SQ(OP)")
b = src.index("#undef WT_SCAN_LANE")
op, df = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
text = (head + "\n#define WT_MARK(x) do { } while (0)\n#define WT_TICK(slot) do { } while (0)\n" + src[a:b] +
        "\ntemplate __global__ void wt_delta_kernel<%s, %s>(const WtParams);\n" % (op, "true" if df != "0" else "false"))
d = os.path.join(ROOT, "wiggletools_amd/csrc")
with tempfile.NamedTemporaryFile("w", suffix=".hip", dir=d, delete=False) as f:
    f.write(text)
    name = f.name
try:
    out = name[:-4] + ".s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-Wno-pass-failed",
                           "--cuda-device-only", "-S", name, "-o", out] + flags)
    sys.stdout.write(open(out).read())
    os.unlink(out)
finally:
    os.unlink(name)
