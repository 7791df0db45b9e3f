"""Summarises a rocprofv3 --kernel-trace --memory-copy-trace run of tools/e2e_only.py: how much of
the host->device gather, the multiplex/reduce kernels and the device->host result copies of the
streaming pipeline (csrc/wt_pipe.h) ran at the same time.  usage: pipe_overlap.py <trace dir> <out.json>"""
import csv
import glob
import json
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(u):
    return sum(b - a for a, b in u)


def inter(u, v):
    i = j = 0
    t = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b:
            t += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return t


def main(d, out):
    gather, comp, export, d2h, h2d = [], [], [], [], []
    names = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "wt_" not in n:
                continue
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            key = n.split("(")[0][:40]
            names.setdefault(key, [0, 0])
            names[key][0] += 1
            names[key][1] += b - a
            (gather if "wt_gather" in n else (export if "wt_export" in n else comp)).append((a, b))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            direction = r.get("Direction", "")
            if "DEVICE_TO_HOST" in direction:
                d2h.append((a, b))
            elif "HOST_TO_DEVICE" in direction:
                h2d.append((a, b))
    if not comp:
        print("no wt_ kernels in", d)
        return
    # only the pipeline's span (the bulk leg): from the first gather kernel to the last of them + tail
    t0 = min(a for a, b in gather) if gather else min(a for a, b in comp)
    t1 = max(b for a, b in gather) if gather else max(b for a, b in comp)
    clip = lambda iv: [(max(a, t0), min(b, t1 + 5_000_000)) for a, b in iv if b > t0 and a < t1 + 5_000_000]
    ug, uc, ud = union(clip(gather)), union(clip(comp)), union(clip(export))
    # steady state: the second half of the gather launches (buffers have stopped growing)
    gs = sorted(gather)
    half = gs[len(gs) // 2:]
    steady_span = half[-1][1] - half[0][0] if len(half) > 1 else 0
    steady_busy = sum(b - a for a, b in half)
    span = (t1 - t0)
    res = {"what": "rocprofv3 --kernel-trace of tools/e2e_only.py: H2D = wt_gather_kernel (copy stream), compute = wt_index + wt_delta (+ wt_reduce for the Multiplexer's priming batch), D2H = wt_export_kernel (result stream)",
           "span_ms": span / 1e6,
           "h2d_gather_busy_ms": total(ug) / 1e6, "compute_busy_ms": total(uc) / 1e6, "d2h_export_busy_ms": total(ud) / 1e6,
           "compute_under_h2d_ms": inter(ug, uc) / 1e6, "d2h_under_h2d_ms": inter(ug, ud) / 1e6,
           "frac_of_compute_hidden_under_h2d": inter(ug, uc) / max(total(uc), 1),
           "frac_of_d2h_hidden_under_h2d": inter(ug, ud) / max(total(ud), 1),
           "h2d_link_busy_frac_of_span": total(ug) / max(span, 1),
           "steady_state_h2d_link_busy_frac": steady_busy / max(steady_span, 1),
           "kernels": {k: {"launches": v[0], "total_ms": v[1] / 1e6} for k, v in names.items()},
           "n_d2h_copies": len(d2h), "n_h2d_copies": len(h2d)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
