"""Fills the @@NAME@@ placeholders of the assembled DESIGN.md from profiles/r05_bench_default.json and friends.
Usage: python tools/fill_design.py /tmp/DESIGN_new.md > DESIGN.md"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))


def sci(x, digits=2):
    if x is None:
        return "n/a"
    e = 0
    m = float(x)
    while abs(m) >= 10:
        m /= 10; e += 1
    while 0 < abs(m) < 1:
        m *= 10; e -= 1
    sup = str.maketrans("0123456789-", "⁰¹²³⁴⁵⁶⁷⁸⁹⁻")
    return ("%." + str(digits) + "f") % m + "·10" + str(e).translate(sup)


cfg, rf = d["config"], d["roofline"]
g = d["e2e_bigwig_genome"]
fp = g.get("fresh_process") or {}
subs, runs, others = d["configs"], d["c2_runs"], d["other_kernels"]
cb = d["cpu_baseline"]
lv = d.get("e2e_bigwig_levels", {})
e2e = d.get("e2e", {})
launches = 24
V = {
    "C2_KERNEL_MS": "%.1f" % rf["kernel_ms"], "C2_INDEX_MS": "%.1f" % rf["index_kernel_ms"], "C2_STEP_MS": "%.1f" % d["ms_per_step"],
    "C2_FRAC": "%.3f" % rf["frac"], "C2_FRAC_IDX": "%.3f" % rf["frac_with_index"], "C2_VALUE": sci(d["value"]),
    "C2_ROCPROF_MS": "%.2f" % tr["kernel_ms_rocprof_avg"], "C2_ROCPROF_SUM": "%.1f" % (tr["kernel_ms_rocprof_avg"] * launches),
    "C2_TRAFFIC_RATIO": "%.3f" % tr["hbm_bytes_per_algorithmic_byte"], "C2_TRAFFIC_GB": "%.2f" % (tr["hbm_bytes_per_launch"] / 1e9),
    "C2_ALG_GB": "%.2f" % (tr["algorithmic_bytes_per_launch_of_that_run"] / 1e9),
    "L1_FRAC": "%.3f" % runs["l1"]["roofline"]["frac"], "L200_FRAC": "%.3f" % runs["l200"]["roofline"]["frac"],
    "FULLM_FRAC": "%.3f" % runs["full_mantissa"]["roofline"]["frac"], "PATCHED_FRAC": "%.3f" % runs["full_mantissa_patched"]["roofline"]["frac"],
    "SUM500_FRAC": "%.3f" % others["sum_500"]["roofline"]["frac"], "MAX_FRAC": "%.3f" % others["max"]["roofline"]["frac"],
    "PRODUCT_FRAC": "%.3f" % others["product"]["roofline"]["frac"], "F64_FRAC": "%.3f" % others["mean_f64_values"]["roofline"]["frac"],
    "C3_MS": "%.1f" % subs["c3"]["ms_per_step"], "C3_VALUE": sci(subs["c3"]["value"]), "C3_FRAC": "%.3f" % subs["c3"]["roofline"]["frac"],
    "C4_MS": "%.0f" % subs["c4"]["ms_per_step"], "C4_VALUE": sci(subs["c4"]["value"]), "C4_FRAC": "%.3f" % subs["c4"]["roofline"]["frac"],
    "C5_MS": "%.0f" % subs["c5"]["ms_per_step"], "C5_VALUE": sci(subs["c5"]["value"]), "C5_FRAC": "%.4f" % subs["c5"]["roofline"]["frac"],
    "C5_S": "%.2f" % (subs["c5"]["ms_per_step"] / 1e3),
    "TT_MS": "%.0f" % subs["ttest"]["ms_per_step"], "TT_VALUE": sci(subs["ttest"]["value"]), "TT_FRAC": "%.3f" % subs["ttest"]["roofline"]["frac"],
    "E2E_COLD": sci(g["bp_per_s"]), "E2E_COLD_S": "%.2f" % g["cold"]["seconds"], "E2E_WARM": sci(g["warm_bp_per_s"]), "E2E_WARM_S": "%.2f" % g["warm"]["seconds"],
    "E2E_STEADY": sci(g["steady_bp_per_s"]), "E2E_DECODE_MS": "%.2f" % (g["warm"]["sum_device_decode_ms"] / 1e3),
    "E2E_FRESH": sci(fp.get("bp_per_s")), "E2E_FRESH_S": "%.2f" % fp.get("seconds", float("nan")), "E2E_FRESH_LOAD": "%.3f" % fp.get("library_load_s", float("nan")),
    "Z1_STEADY": sci(lv.get("z1", {}).get("steady_bp_per_s")), "Z6_STEADY": sci(lv.get("z6", {}).get("steady_bp_per_s")),
    "E2E_BULK": sci((e2e.get("bulk") or {}).get("bp_per_s")), "E2E_BULK_STEADY": sci((e2e.get("bulk") or {}).get("steady_bp_per_s")),
    "E2E_POP": sci((e2e.get("pop") or {}).get("bp_per_s")), "E2E_BUFFERED": sci((e2e.get("buffered") or {}).get("bp_per_s")),
    "CPU1": sci(cb["value"]), "CPU16": sci(cb.get("many_core", {}).get("value")),
    "C2_X_CPU1": sci(d["value"] / cb["value"], 1), "C2_X_CPU16": sci(d["value"] / cb["many_core"]["value"], 1),
    "E2E_X_CPU1": "%.0f" % (g["bp_per_s"] / cb["value"]), "E2E_X_CPU16": "%.0f" % (g["bp_per_s"] / cb["many_core"]["value"]),
    "BENCH_S": "%.0f s" % d["bench_seconds"],
    "GPU_TESTS": open(os.path.join(ROOT, "profiles", "r05_gpu_tests.log")).read().strip().splitlines()[-1].strip(),
    "CPU_TESTS": os.environ.get("WT_CPU_TESTS", "see the round's CPU run"),
}
text = open(sys.argv[1]).read()
missing = set(re.findall(r"@@([A-Z0-9_]+)@@", text)) - set(V)
if missing:
    sys.stderr.write("unfilled: %s\n" % sorted(missing))
text = re.sub(r"@@([A-Z0-9_]+)@@", lambda m: V.get(m.group(1), m.group(0)), text)
# the drop-in layer's file was split this round
text = text.replace("Host C++ (`csrc/wt_iter_abi.cpp`), because", "Host C++ (`csrc/wt_iter_abi.cpp` + the `csrc/wt_abi_*.h` it includes: common, feeder, reduce, readers, bwdev, ops, integrators -- one translation unit), because")
# §4.2's description was written in round 2: say so where the plan has changed since
text = text.replace("Window: 4096 bp, 512 lanes, ONE 142 KB workgroup (8 waves) per CU.  The first version",
                    "Window: 4096 bp; rounds 2-4: 512 lanes, ONE 142 KB workgroup (8 waves) per CU (round 5: 1024 lanes, 147 KB, 16 waves -- \"Today\" below).  The first version")
text = text.replace("C3 on MI355X: var + stddev over 500 tracks of chromosome 1 in 100 ms (both reducers, index once):",
                    "C3 on MI355X in round 2: var + stddev over 500 tracks of chromosome 1 in 100 ms (both reducers, index once):")
text = text.replace("CU; rounds 1-2 and the launches with squares: T = 512, W = 4096, 66 KB, 2 workgroups / CU):",
                    "CU; rounds 1-2: T = 512, W = 4096, 66 KB, 2 workgroups / CU; the launches with squares: W = 4096, 146 KB, one workgroup of 768 lanes of which the first 512 run steps 3-5, §4.2):")
sys.stdout.write(text)
