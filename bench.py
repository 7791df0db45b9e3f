#!/usr/bin/env python
"""bench.py -- throughput of the hot path (Multiplexer -> reducer) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1
launched by torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs), `--config`:
  c2 (default)  mean over 100 synthetic tracks, the WHOLE 3.1 Gbp genome (24 chromosomes, GRCh38 lengths)
  c3            var + stddev over 500 tracks, chromosome 1 (248 Mbp) -- a step runs both reducers
  c4            median over 100 tracks, whole genome
  c5            wilcoxon 50 v 50, whole genome (+ the scalar gathers: AUC and Pearson)
Tracks: float32 values k/8, run length ~ 1 + Geometric(1/l) (l = --mean-run, default 16), 2 % gaps,
from the counter-based generator csrc/wt_synth.hip.

One STEP = one pass of the hot path over the whole genome of the configuration.  100 tracks x
3.1 Gbp at l = 16 are 233 GB of run lists -- they do not fit one GPU together with the output -- so a
pass walks a host-side work queue of chromosomes (largest first): generate the chromosome's
tracks in HBM (NOT timed), then window-index kernel + fused multiplex/reduce kernel over it with
inputs and outputs resident (timed: synchronise, clock, launch, synchronise, clock).  ms_per_step
is the sum of the timed sections of one pass; bp/s = covered bp / that time.

Multi-GPU (--shard genome, default): ONE genome per step, its chromosomes handed out to the
ranks by a shared work queue (a counter in the torch.distributed store) -- strong scaling, no
collective on the data path; per-pass time = max over ranks.  Genome-wide scalars travel through
RCCL after the timed region: AUC / bp / run counts by all_reduce, the Pearson moments of tracks
0,1 by all_gather + ordered pairwise merge (reference statistics.c:442-456 is sequential).
--shard replicas: every rank walks its own genome (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 20260927

CONFIGS = {
    "c2": dict(ops=["mean"], tracks=100, chroms=list(range(24)), what="mean over 100 tracks, whole genome"),
    "c3": dict(ops=["var", "stddev"], tracks=500, chroms=[0], what="var + stddev over 500 tracks, chromosome 1"),
    "c4": dict(ops=["median"], tracks=100, chroms=list(range(24)), what="median over 100 tracks, whole genome"),
    "c5": dict(ops=["wilcoxon"], tracks=100, chroms=list(range(24)), what="wilcoxon 50 v 50, whole genome"),
}


# ---------------------------------------------------------------------------------------------
# CPU baselines (the checker's compiled reference, oracle/_ref; else the C restatement)
# ---------------------------------------------------------------------------------------------
def _sample_tracks(chrom_id, chrom_len, n_tracks, mean_run, lo, hi):
    """The bench's own tracks of one chromosome, run starts in [lo, hi), finishes clipped to hi: the
    counter-based generator's numpy mirror regenerates them on the host (nothing is copied back)."""
    from wiggletools_amd import synthgen
    t = synthgen.host_runlists(SEED, [chrom_len], n_tracks, mean_run, 0.02, 800, region=(lo, hi), chrom_ids=[chrom_id])
    d = t.as_dict()
    d["finish"] = np.minimum(d["finish"], hi + 1).astype(np.int32)
    return d


def effective_cores():
    """Host cores this process may actually use: min(cpu_count, scheduler affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return n


def _sample_tracks_device(chrom_id, chrom_len, n_tracks, mean_run, hi):
    """Same sample (run starts in [0, hi) of one chromosome) taken from the device generator: the
    single-thread leg's sample is tens of Mbp, which the numpy mirror would take longer to hash than
    the reference takes to evaluate."""
    import torch
    from wiggletools_amd import synthgen
    seg, s, f, v = synthgen.device_tracks(SEED, [chrom_len], n_tracks, mean_run, 0.02, 800, chrom_ids=[chrom_id])
    S, F, V, so = [], [], [], [0]
    for t in range(n_tracks):
        a, b = int(seg[t]), int(seg[t + 1])
        k = int(torch.searchsorted(s[a:b], torch.tensor([hi], device=s.device, dtype=s.dtype), right=True).item())   # 1-based start <= hi
        S.append(s[a:a + k].cpu().numpy())
        F.append(np.minimum(f[a:a + k].cpu().numpy(), hi + 1))
        V.append(v[a:a + k].cpu().numpy().astype(np.float64))
        so.append(so[-1] + k)
    del s, f, v
    return dict(n_chrom=1, n_tracks=n_tracks, seg_off=np.array(so, np.int64), start=np.concatenate(S),
                finish=np.concatenate(F).astype(np.int32), value=np.concatenate(V), defaults=np.zeros(n_tracks))


def _time_cpu(d, op, n_set0):
    from oracle import oracle as O
    O.build()
    code = O.OPS["mwu" if op == "wilcoxon" else op]
    if O.have_ref() and code <= 9:
        sec, runs, bp = O.ref_time_reduce(d, code)
        return sec, bp, "reference"
    t0 = time.perf_counter()
    c, s, f, v = O.reduce(d, code, n_set0=n_set0)
    return time.perf_counter() - t0, int((f.astype(np.int64) - s).sum()), "port"


def _many_core_worker(args):
    (k, chrom_id, chrom_len, n_tracks, mean_run, lo, hi, op, n_set0, t_go) = args
    d = _sample_tracks(chrom_id, chrom_len, n_tracks, mean_run, lo, hi)
    while time.time() < t_go:           # common start: the regions are evaluated side by side
        time.sleep(0.001)
    t0 = time.time()
    sec, bp, kind = _time_cpu(d, op, n_set0)
    return t0, time.time(), bp, kind


def cpu_baseline(chrom_ids, op, n_tracks, mean_run, chrom_lens, many_core=True, target_s=12.0):
    """(a) as shipped: ONE evaluation thread (the reference never parallelises evaluation);
    (b) reference-style many-core: one process per 30 Mbp region on every host core
    (reference python/wiggletools/parallelWiggleTools.py:63,109), each timing a bounded slice of
    its region.  Both on the bench's own tracks of the largest chromosome, sink = none."""
    n_set0 = n_tracks // 2 if op in ("wilcoxon", "mwu", "ttest") else 0
    k0 = int(np.argmax(chrom_lens))
    c0, clen = int(chrom_ids[k0]), int(chrom_lens[k0])
    probe = _sample_tracks(c0, clen, n_tracks, mean_run, 0, min(100_000, clen))
    sec, bp, kind = _time_cpu(probe, op, n_set0)
    rate = bp / max(sec, 1e-9)
    sample_bp = int(min(max(rate * target_s, 100_000), clen, 16_000_000))
    sec, bp, kind = _time_cpu(_sample_tracks_device(c0, clen, n_tracks, mean_run, sample_bp), op, n_set0)
    out = {"value": bp / sec, "unit": "genomic bp/s", "cores": 1, "kind": kind,
           "sample": "%s over the bench's own %d tracks, first %d bp of the largest chromosome, one evaluation thread "
                     "(the reference never parallelises evaluation), sink=none; %.1f s" % (op, n_tracks, sample_bp, sec)}
    nproc = effective_cores()
    out["host_cores"] = nproc
    out["host_cores_note"] = "os.cpu_count() %d, cgroup / affinity limit %d" % (os.cpu_count() or 1, nproc)
    if many_core and nproc > 1:
        import multiprocessing as mp
        region = 30_000_000
        workers = min(nproc, 256)
        slice_bp = int(min(max(out["value"] * 4.0, 50_000), 2_000_000))     # ~4 s of evaluation per worker
        # every worker first regenerates its slice with the generator's numpy mirror (~40 ns per
        # track position), then all start evaluating at t_go
        t_go = time.time() + 10.0 + slice_bp * n_tracks * 6e-8
        jobs = []
        for k in range(workers):
            lo = (k * region) % max(clen - slice_bp, 1)
            jobs.append((k, c0, clen, n_tracks, mean_run, lo, lo + slice_bp, op, n_set0, t_go))
        try:
            with mp.get_context("spawn").Pool(workers) as pool:
                res = pool.map(_many_core_worker, jobs, chunksize=1)
            t_first = min(r[0] for r in res)
            t_last = max(r[1] for r in res)
            late = max(r[0] for r in res) - t_go
            out["many_core"] = {"value": sum(r[2] for r in res) / (t_last - t_first), "unit": "genomic bp/s",
                                "cores": nproc, "workers": workers, "kind": res[0][3],
                                "sample": "one process per 30 Mbp region (parallelWiggleTools.py:63,109), %d regions side by side, "
                                          "each evaluating the first %d bp of its region; wall %.1f s, latest start %+.2f s after the common go"
                                          % (workers, slice_bp, t_last - t_first, late)}
        except Exception as e:      # never lose the bench line to the baseline
            out["many_core"] = {"error": repr(e)[:200], "cores": nproc}
    return out


# ---------------------------------------------------------------------------------------------
# End to end through the drop-in layer: first pop -> last run on the host (SURVEY 8d metric 1)
# ---------------------------------------------------------------------------------------------
def e2e_dropin(op, n_tracks, mean_run, mbp, device):
    """The reference's own C API (newMultiplexer + <op>Reduction) over N array-backed tracks in
    PINNED host memory, from the constructor (which primes: the first pop) to the last run on the
    host.  Two legs on the same tracks:
      pop   the reference's protocol on both sides: children popped one interval per indirect call (by a
            few worker threads, each child always by the same one), runs taken one pop at a time (what
            any foreign reader / consumer gets);
      bulk  the library's bulk doors: children hand over SoA blocks that the copy engine reads where
            they lie, runs taken in blocks.
    Both run the pinned-staging / 3-stream pipeline of csrc/wt_pipe.h underneath."""
    import torch
    from wiggletools_amd import dropin, synthgen
    L = int(mbp * 1e6)
    seg, s, f, v = synthgen.device_tracks(SEED, [L], n_tracks, mean_run, 0.02, 800, device, chrom_ids=[40])
    n = int(seg[-1])
    hs, hf, hv = dropin.PinnedArray(n, np.int32), dropin.PinnedArray(n, np.int32), dropin.PinnedArray(n, np.float32)
    torch.from_numpy(hs.array).copy_(s); torch.from_numpy(hf.array).copy_(f); torch.from_numpy(hv.array).copy_(v)
    torch.cuda.synchronize()
    del s, f, v
    torch.cuda.empty_cache()

    def readers(limit_bp):
        its = []
        for t in range(n_tracks):
            a, b = int(seg[t]), int(seg[t + 1])
            if limit_bp < L:
                b = a + int(np.searchsorted(hs.array[a:b], limit_bp))
            its.append(dropin.array_reader(["chr1"], [0, b - a], hs.ptr + 4 * a, hf.ptr + 4 * a, hv.ptr + 4 * a))
        return its

    out = {"tracks": n_tracks, "mean_run_bp": mean_run, "op": op, "host_bytes_per_bp": 12.0 * n / L,
           "pcie_h2d_roofline_bp_per_s": 63e9 / (12.0 * n / L)}
    # bulk leg
    os.environ.pop("WTAMD_NO_BULK", None)
    t0 = time.perf_counter()
    r = dropin.reducer(op, readers(L), n_set0=n_tracks // 2)
    marks = []
    runs, _ = dropin.drain_blocks(r, on_block=lambda c, a, b, v: (marks.append((time.perf_counter(), int(b[-1]))), 0)[1])
    dt = time.perf_counter() - t0
    st = dropin.pipe_stats(r)
    out["bulk"] = {"bp_per_s": L / dt, "seconds": dt, "bp": L, "runs": runs,
                   "h2d_GBs": 12.0 * n / dt / 1e9, "d2h_GBs": 16.0 * runs / dt / 1e9,
                   "batches": st.get("batches"), "sum_h2d_ms": st.get("h2d_ms"), "sum_kernel_ms": st.get("kernel_ms"),
                   "sum_d2h_ms": st.get("d2h_ms"), "host_submit_ms": st.get("host_submit_ms"), "host_wait_ms": st.get("host_wait_ms")}
    # steady state: from the block that ends the first quarter to the last (buffers have stopped growing)
    q = [m for m in marks if m[1] >= L // 4]
    if len(q) >= 2 and q[-1][0] > q[0][0]:
        out["bulk"]["steady_bp_per_s"] = (q[-1][1] - q[0][1]) / (q[-1][0] - q[0][0])
    # pop legs (a slice: they are slower): children drained by the library's worker threads (a child always by
    # the same thread; the default for 16 or more foreign children) and by ONE thread.  The slice is COMPACTED first
    # (the first 20 Mbp of every track, back to back): a per-interval pop() walks 100 streams at once, and with every
    # stream 190 MB from the next (chromosome-sized arrays) it measured the host's TLB, not the protocol (1.4e7 vs
    # 3.2e7 bp/s); the reference's readers hand over compact 10 000-entry blocks (bufferedReader.c:21-28).
    os.environ["WTAMD_NO_BULK"] = "1"
    pop_bp = int(min(L, 20e6))
    cuts = [int(seg[t]) + int(np.searchsorted(hs.array[int(seg[t]):int(seg[t + 1])], pop_bp)) for t in range(n_tracks)]
    pseg = np.concatenate([[0], np.cumsum([cuts[t] - int(seg[t]) for t in range(n_tracks)])]).astype(np.int64)
    ps, pf, pv = dropin.PinnedArray(int(pseg[-1]), np.int32), dropin.PinnedArray(int(pseg[-1]), np.int32), dropin.PinnedArray(int(pseg[-1]), np.float32)
    for t in range(n_tracks):
        a, b, o = int(seg[t]), cuts[t], int(pseg[t])
        ps.array[o:o + b - a] = hs.array[a:b]; pf.array[o:o + b - a] = hf.array[a:b]; pv.array[o:o + b - a] = hv.array[a:b]

    def pop_readers():
        return [dropin.array_reader(["chr1"], [0, int(pseg[t + 1] - pseg[t])], ps.ptr + 4 * int(pseg[t]), pf.ptr + 4 * int(pseg[t]),
                                    pv.ptr + 4 * int(pseg[t])) for t in range(n_tracks)]

    for key, threads in (("pop", None), ("pop_one_drain_thread", "1")):
        if threads is None:
            os.environ.pop("WTAMD_DRAIN_THREADS", None)
        else:
            os.environ["WTAMD_DRAIN_THREADS"] = threads
        t0 = time.perf_counter()
        r = dropin.reducer(op, pop_readers(), n_set0=n_tracks // 2)
        runs, bp, acc = dropin.drain_pops(r)
        dt = time.perf_counter() - t0
        out[key] = {"bp_per_s": bp / dt, "seconds": dt, "bp": bp, "runs": runs,
                    "child_pops_per_s": float(pseg[-1]) / dt,
                    "drain_threads": threads or ("auto: min(16, usable cores = %d) for >= 16 foreign children" % effective_cores())}
    # the same slice behind readers built on the buffered reader (csrc/wt_bufreader.h, the reference's bufferedReader.c
    # replaced): a producer thread per track pushes one interval at a time, the Multiplexer takes the 10 000-entry blocks
    # whole.  What an unchanged reference reader (bigWiggleReader.c, bamReader.c ...) gets once it links this library.
    os.environ.pop("WTAMD_NO_BULK", None)
    os.environ.pop("WTAMD_DRAIN_THREADS", None)
    t0 = time.perf_counter()
    r = dropin.reducer(op, [dropin.buffered_array_reader(["chr1"], [0, int(pseg[t + 1] - pseg[t])], ps.ptr + 4 * int(pseg[t]),
                                                         pf.ptr + 4 * int(pseg[t]), pv.ptr + 4 * int(pseg[t])) for t in range(n_tracks)],
                       n_set0=n_tracks // 2)
    runs, _ = dropin.drain_blocks(r)
    dt = time.perf_counter() - t0
    out["buffered"] = {"bp_per_s": pop_bp / dt, "seconds": dt, "bp": pop_bp, "runs": runs, "child_entries_per_s": float(pseg[-1]) / dt,
                       "note": "children = producer threads pushing into the buffered reader (one call per interval); blocks taken whole"}
    ps.free(); pf.free(); pv.free()
    hs.free(); hf.free(); hv.free()
    return out


def trim_pools():
    """The library keeps a finished reducer's staging and device buffers for the next one (csrc/wt_pipe.h); the bench's
    other records allocate through torch and want that memory back."""
    from wiggletools_amd import _lib
    _lib.lib().wtamd_pool_trim()


def _bw_write_one(args):
    from wiggletools_amd import bwwrite
    path, L, s, f, v = args
    return bwwrite.write_arrays(path, {"chr1": L}, {"chr1": (s - 1, f - 1, v)})


def e2e_bigwig(op, n_tracks, mean_run, mbp, device):
    """FILES to result: N BigWig files (written here, untimed, from the same generator; bedGraph sections of 1024
    items, zlib level 1 -- SURVEY 8d's stored form) -> wtamd_BigWiggleReader x N -> newMultiplexer -> <op>Reduction ->
    runs on the host.  What `wiggletools <op> *.bw` does in the reference (commandParser.c -> bigWiggleReader.c ->
    multiplexer.c -> reducers.c), minus the text writer.  The sections travel to the GPU COMPRESSED and are inflated
    and decoded there (csrc/wt_bwdev.hip); `host_decoder` times the library's host-side zlib route on a seek window of
    the same files for comparison."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from wiggletools_amd import dropin, synthgen
    L = int(mbp * 1e6)
    keep = os.environ.get("WTAMD_BENCH_BWDIR")          # experiments: files written once, reused by later runs
    meta = os.path.join(keep, "meta_%d_%d.json" % (n_tracks, L)) if keep else None
    if keep and os.path.exists(meta):
        d, write_s = keep, 0.0
        n_int = json.load(open(meta))["intervals"]
        jobs = [(os.path.join(d, "t%03d.bw" % t),) for t in range(n_tracks)]
    else:
        seg, s, f, v = synthgen.device_tracks(SEED, [L], n_tracks, mean_run, 0.02, 800, device, chrom_ids=[41])
        hs, hf, hv = s.cpu().numpy(), f.cpu().numpy(), v.cpu().numpy()
        del s, f, v
        base = os.environ.get("WTAMD_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 6.0 * len(hs) else None)
        if keep:
            os.makedirs(keep, exist_ok=True)
            d = keep
        else:
            d = tempfile.mkdtemp(prefix="wtamd_bw_", dir=base)
        jobs = [(os.path.join(d, "t%03d.bw" % t), L + 1, hs[int(seg[t]):int(seg[t + 1])], hf[int(seg[t]):int(seg[t + 1])],
                 hv[int(seg[t]):int(seg[t + 1])]) for t in range(n_tracks)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max(1, min(effective_cores(), 16))) as ex:     # zlib releases the GIL
            list(ex.map(_bw_write_one, jobs))
        write_s = time.perf_counter() - t0
        del hs, hf, hv
        n_int = int(seg[-1])
        if keep:
            json.dump({"intervals": n_int}, open(meta, "w"))
    try:
        size = sum(os.path.getsize(j[0]) for j in jobs)
        os.environ.pop("WTAMD_BW_DEVICE", None)

        def pool():
            import ctypes as C
            from wiggletools_amd import _lib
            a = (C.c_int64 * 6)()
            _lib.lib().wtamd_pool_stats(a)
            return list(a)

        def run_once():
            p0 = pool()
            t0 = time.perf_counter()
            readers = dropin.bigwig_readers([j[0] for j in jobs], box=True)
            t_readers = time.perf_counter() - t0
            r = dropin.reducer(op, readers, n_set0=n_tracks // 2)
            t_open = time.perf_counter() - t0
            marks = []
            runs, _ = dropin.drain_blocks(r, on_block=lambda c, a, b, v: (marks.append((time.perf_counter(), int(b[-1]))), 0)[1])
            dt = time.perf_counter() - t0
            st = dropin.pipe_stats(r)
            o = {"seconds": dt, "bp_per_s": L / dt, "runs": runs, "intervals_per_s": n_int / dt, "inbound_GBs": size / dt / 1e9,
                 "open_seconds": t_open, "open_readers_seconds": t_readers, "batches": st.get("batches"),
                 "sections_inflated_on_device": st.get("bw_sections"), "sum_device_decode_ms": st.get("bw_decode_ms"),
                 "sum_kernel_ms": st.get("kernel_ms"), "sum_d2h_ms": st.get("d2h_ms"), "host_submit_ms": st.get("host_submit_ms"),
                 "host_wait_ms": st.get("host_wait_ms")}
            p1 = pool()
            # buffers of 1 MB and more this run had to get from the runtime instead of the process-wide pools
            o["pinned_afresh"] = {"buffers": p1[0] - p0[0], "bytes": p1[1] - p0[1]}     # hipHostMalloc
            o["device_afresh"] = {"buffers": p1[3] - p0[3], "bytes": p1[4] - p0[4]}     # hipMalloc
            q = [m for m in marks if m[1] >= L // 4]        # ramp-up excluded: from the block ending the first quarter on
            if len(q) >= 2 and q[-1][0] > q[0][0]:
                o["steady_bp_per_s"] = (q[-1][1] - q[0][1]) / (q[-1][0] - q[0][0])
            return o

        cold = run_once()       # the process's first pipe: hardware queues are created, ~1.7 GB of staging is pinned
        warm = run_once()       # a long-lived process: queues exist, the pinned pool holds the staging
        out = {"tracks": n_tracks, "op": op, "bp": L, "intervals": n_int, "file_bytes": size, "file_bytes_per_bp": size / L,
               "pcie_h2d_roofline_bp_per_s": 63e9 / (size / L), "files_written_s": write_s, "files_dir": d.rsplit("/", 1)[0],
               "host_cores": effective_cores(),
               "decoder": "device (one lane per zlib stream)" if (cold.get("sections_inflated_on_device") or 0) > 0 else "host zlib",
               "timed": "open 100 files (index walk, priming) -> newMultiplexer -> MeanReduction -> every run on the host",
               # `bp_per_s` = the cold run, everything included; `warm` = the same again in this process; `steady` = ramp-up excluded
               "seconds": cold["seconds"], "bp_per_s": cold["bp_per_s"], "runs": cold["runs"],
               "warm_bp_per_s": warm["bp_per_s"], "steady_bp_per_s": warm.get("steady_bp_per_s") or cold.get("steady_bp_per_s"),
               "cold": cold, "warm": warm}
        # the host-side decoder (round 2's route) on a window of the same files
        if os.environ.get("WTAMD_BENCH_NO_HOSTDEC"):
            return out
        try:
            win = int(min(L, 8e6))
            os.environ["WTAMD_BW_DEVICE"] = "0"
            t0 = time.perf_counter()
            r = dropin.reducer(op, [dropin.bigwig_reader(j[0], box=True) for j in jobs], n_set0=n_tracks // 2)
            dropin.seek(r, "chr1", 1, win + 1)
            runs_h, _ = dropin.drain_blocks(r)
            dth = time.perf_counter() - t0
            out["host_decoder"] = {"bp": win, "seconds": dth, "bp_per_s": win / dth, "runs": runs_h,
                                   "note": "WTAMD_BW_DEVICE=0: zlib inflate + section decode on one host thread per file, seek window chr1:1-%d" % win}
        except Exception as e:
            out["host_decoder"] = {"error": repr(e)[:200]}
        finally:
            os.environ.pop("WTAMD_BW_DEVICE", None)
        return out
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def chrom_name(c):
    return "chr%d" % (c + 1) if c < 22 else ("chrX" if c == 22 else "chrY")


def e2e_bigwig_genome(op, n_tracks, mean_run, scale, device, only=None, level=1, runs=2, fresh_process=False):
    """The north-star measurement: `op` over n_tracks WHOLE-GENOME BigWig files -> result on the host.  Every file holds
    all 24 chromosomes (GRCh38 lengths x `scale`, names chr1 .. chr22, chrX, chrY; the reader walks them in strcmp
    order as reference src/bigWiggleReader.c:91-101 does, 10 000-bp stretches :52-83), bedGraph sections of 1024 items,
    zlib level 1 (SURVEY 8d's stored form), written here untimed by the library's native writer (csrc/wt_bwwrite.cpp)
    from the same counter-based generator as the resident legs.  Timed: open the files (index walk, priming) ->
    newMultiplexer -> <op>Reduction -> every run on the host -- `wiggletools <op> *.bw` minus the text writer.  cold = the
    process's first such run (hardware queues, pinned staging and device buffers obtained afresh), warm = the same again
    (pools filled), steady = warm run from the first quarter of the genome on."""
    import shutil
    import tempfile
    from wiggletools_amd import bwwrite, dropin, synthgen
    import torch
    lens = [max(int(g * scale), 1000) for g in GRCH38]
    if only is not None:                # experiments: data on these chromosomes only (the others stay in the files' trees, empty)
        lens = [lens[c] if c in only else 0 for c in range(24)]
    names = [chrom_name(c) for c in range(24)]
    order = sorted(range(24), key=lambda c: names[c].encode())          # strcmp order = id order inside the files
    genome_bp = sum(lens)
    keep = os.environ.get("WTAMD_BENCH_BWDIR")
    d = keep or tempfile.mkdtemp(prefix="wtamd_bwg_", dir=os.environ.get("WTAMD_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
    os.makedirs(d, exist_ok=True)
    paths = [os.path.join(d, "g%03d.bw" % t) for t in range(n_tracks)]
    meta = os.path.join(d, "meta_genome_%d_%d_%s_z%d.json" % (n_tracks, genome_bp, "all" if only is None else "-".join(map(str, sorted(only))), level))
    try:
        if keep and os.path.exists(meta):
            m = json.load(open(meta))
            n_int, write_s, gen_s, sections = m["intervals"], 0.0, 0.0, m["sections"]
        else:
            t0 = time.perf_counter()
            fs = bwwrite.FileSet(paths, {names[c]: lens[c] + 1 for c in range(24)}, items_per_block=1024, level=level, threads=max(1, min(effective_cores(), 32)))
            n_int, gen_s = 0, 0.0
            for c in order:
                if lens[c] == 0:
                    continue
                g0 = time.perf_counter()
                seg, s_, f_, v_ = synthgen.device_tracks(SEED, [lens[c]], n_tracks, mean_run, 0.02, 800, device, chrom_ids=[c])
                hs, hf, hv = s_.cpu().numpy(), f_.cpu().numpy(), v_.cpu().numpy()
                del s_, f_, v_
                gen_s += time.perf_counter() - g0
                fs.add_chrom(names[c], seg, hs, hf, hv)
                n_int += int(seg[-1])
                del hs, hf, hv
            sections = int(sum(fs.close()))
            write_s = time.perf_counter() - t0
            torch.cuda.empty_cache()
            if keep:
                json.dump({"intervals": n_int, "sections": sections}, open(meta, "w"))
        size = sum(os.path.getsize(p_) for p_ in paths)

        def fresh_run():
            """`wiggletools <op> *.bw` as a process of its own (tools/cli_cold.py): nothing of this process's state helps it."""
            import subprocess
            try:
                env = {k: v for k, v in os.environ.items() if not k.startswith("WTAMD_TRACE")}
                pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_cold.py"), d, str(n_tracks), op, str(genome_bp)],
                                    capture_output=True, text=True, timeout=600, env=env)
                lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                return json.loads(lines[-1]) if lines else {"error": (pr.stderr or "no output")[-300:]}
            except Exception as e:
                return {"error": repr(e)[:300]}
        # the very first read of the freshly WRITTEN files (tmpfs pages read at half speed the first time: read_through below) by a
        # fresh process -- reported beside the figures taken after the read-through, which is what round 5 on quotes as cold
        first_touch = fresh_run() if (fresh_process and write_s > 0 and not os.environ.get("WTAMD_BENCH_NO_READ_THROUGH")) else None
        read_through_s = None
        if not os.environ.get("WTAMD_BENCH_NO_READ_THROUGH"):
            read_through_s = read_through(paths)[1]
        os.environ.pop("WTAMD_BW_DEVICE", None)
        starts = {}
        acc = 0
        for c in order:
            starts[names[c].encode()] = acc
            acc += lens[c]

        def pool():
            import ctypes as C
            from wiggletools_amd import _lib
            a = (C.c_int64 * 6)()
            _lib.lib().wtamd_pool_stats(a)
            return list(a)

        def vmstat():
            keys = ("numa_pages_migrated", "numa_hint_faults", "pgmigrate_success", "pgfault", "pgmajfault", "thp_fault_alloc", "pswpin", "pswpout",
                    "pgscan_kswapd", "pgscan_direct", "pgsteal_kswapd", "compact_stall", "thp_collapse_alloc")
            try:
                kv = dict(l.split() for l in open("/proc/vmstat"))
                return {k: int(kv[k]) for k in keys if k in kv}
            except Exception:
                return {}

        def run_once():
            p0 = pool()
            v0 = vmstat() if os.environ.get("WTAMD_BENCH_VMSTAT") else None
            t0 = time.perf_counter()
            readers = dropin.bigwig_readers(paths, box=True)
            t_readers = time.perf_counter() - t0
            r = dropin.reducer(op, readers, n_set0=n_tracks // 2)
            t_open = time.perf_counter() - t0
            marks = []
            seen = set()

            def on_block(c, a, b, v):
                marks.append((time.perf_counter(), starts[c] + int(b[-1])))
                seen.add(c)
                return 0
            runs, _ = dropin.drain_blocks(r, on_block=on_block)
            dt = time.perf_counter() - t0
            st = dropin.pipe_stats(r)
            o = {"seconds": dt, "bp_per_s": genome_bp / dt, "runs": runs, "chromosomes_seen": len(seen), "intervals_per_s": n_int / dt,
                 "inbound_GBs": size / dt / 1e9, "open_seconds": t_open, "open_readers_seconds": t_readers, "batches": st.get("batches"),
                 "sections_inflated_on_device": st.get("bw_sections"), "sum_device_decode_ms": st.get("bw_decode_ms"),
                 "sum_kernel_ms": st.get("kernel_ms"), "sum_d2h_ms": st.get("d2h_ms"), "host_submit_ms": st.get("host_submit_ms"),
                 "host_wait_ms": st.get("host_wait_ms")}
            p1 = pool()
            o["pinned_afresh"] = {"buffers": p1[0] - p0[0], "bytes": p1[1] - p0[1]}
            o["device_afresh"] = {"buffers": p1[3] - p0[3], "bytes": p1[4] - p0[4]}
            if v0 is not None:
                v1 = vmstat()
                o["vmstat_delta"] = {k: v1[k] - v0[k] for k in v0 if v1.get(k, 0) != v0[k]}
            q = [m_ for m_ in marks if m_[1] >= genome_bp // 4]
            if len(q) >= 2 and q[-1][0] > q[0][0]:
                o["steady_bp_per_s"] = (q[-1][1] - q[0][1]) / (q[-1][0] - q[0][0])
            return o

        cold = run_once()
        warm = run_once() if runs > 1 else cold
        # ... and in a FRESH process, which is what a `wiggletools mean *.bw` invocation is: nothing of this process's state
        # (runtime up, code object loaded, queues, pools) helps it (tools/cli_cold.py; the files are still in place)
        fresh = fresh_run() if fresh_process else None
        return {"fresh_process": fresh, "fresh_process_first_touch": first_touch, "files_read_through_s": read_through_s, "tracks": n_tracks, "op": op, "chromosomes_per_file": 24, "genome_scale": scale, "bp": genome_bp, "intervals": n_int, "zlib_level": level,
                "sections": sections, "file_bytes": size, "file_bytes_per_bp": size / genome_bp,
                "pcie_h2d_roofline_bp_per_s": 63e9 / (size / genome_bp), "files_written_s": write_s, "generate_s": gen_s,
                "files_dir": d.rsplit("/", 1)[0], "host_cores": effective_cores(),
                "decoder": "device (one lane per zlib stream)" if (cold.get("sections_inflated_on_device") or 0) > 0 else "host zlib",
                "timed": "open %d files of 24 chromosomes each (index walk, priming) -> newMultiplexer -> %sReduction -> every run on the host" % (n_tracks, op.capitalize()),
                "bp_per_s": cold["bp_per_s"], "warm_bp_per_s": warm["bp_per_s"],
                "steady_bp_per_s": warm.get("steady_bp_per_s") or cold.get("steady_bp_per_s"), "cold": cold, "warm": warm}
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def read_through(paths, threads=16):
    """Reads every file once, untimed.  A tmpfs page that was just WRITTEN is read at half speed the first time (measured:
    19-20 GB/s on the first pass over freshly written files, 40-44 GB/s on every later one, whatever posix_fadvise says --
    the kernel's page-cache bookkeeping of the first access): without this the 'cold' run of the file leg measured the page
    cache's first read of 90 GB (1.0 s of it at full scale: round 5) and not the library's cold start.  Input files of a
    real invocation were written long before and have been read before."""
    from concurrent.futures import ThreadPoolExecutor

    def one(p):
        fd = os.open(p, os.O_RDONLY)
        try:
            buf = bytearray(8 << 20)
            mv = memoryview(buf)
            off = 0
            while True:
                n = os.preadv(fd, [mv], off)
                if n <= 0:
                    break
                off += n
        finally:
            os.close(fd)
        return off
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(threads, effective_cores()))) as ex:
        total = sum(ex.map(one, paths))
    return total, time.perf_counter() - t0


def genome_file_scale(n_tracks, mean_run):
    """Largest genome scale <= 1 (WTAMD_BENCH_GENOME_SCALE) whose file set fits half of what /dev/shm (or WTAMD_BENCH_TMP)
    has free.  (The writing of the files is untimed but not free: 1.9e10 intervals, 90 GB of files, take ~3 minutes of
    zlib on 16 cores at full scale.)"""
    import shutil
    base = os.environ.get("WTAMD_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    try:
        free = shutil.disk_usage(base).free
    except Exception:
        free = 8e9
    per_bp = 0.293 * n_tracks * 16.0 / mean_run        # file bytes per genomic bp: 29.2 measured at 100 tracks, mean run 16
    want = float(os.environ.get("WTAMD_BENCH_GENOME_SCALE", 1.0))
    fit = 0.5 * free / (per_bp * sum(GRCH38))
    return max(min(want, fit), 0.0)


def e2e_sharded(ctx, op, n_tracks, mean_run, mbp):
    """N GPUs, the end-to-end bulk leg per rank: every rank streams its own chromosome (array-backed tracks in pinned
    host memory -> drop-in reducer -> runs on the host) over its own PCIe link; barrier, clock, max over ranks."""
    import torch
    import torch.distributed as dist
    from wiggletools_amd import dropin, synthgen
    L = int(mbp * 1e6 * min(1.0, 100.0 / n_tracks))
    try:
        seg, s, f, v = synthgen.device_tracks(SEED, [L], n_tracks, mean_run, 0.02, 800, ctx.device, chrom_ids=[50 + ctx.rank])
        n = int(seg[-1])
        hs, hf, hv = dropin.PinnedArray(n, np.int32), dropin.PinnedArray(n, np.int32), dropin.PinnedArray(n, np.float32)
        torch.from_numpy(hs.array).copy_(s); torch.from_numpy(hf.array).copy_(f); torch.from_numpy(hv.array).copy_(v)
        torch.cuda.synchronize()
        del s, f, v
        its = []
        for t in range(n_tracks):
            a, b = int(seg[t]), int(seg[t + 1])
            its.append(dropin.array_reader(["chr1"], [0, b - a], hs.ptr + 4 * a, hf.ptr + 4 * a, hv.ptr + 4 * a))
        dist.barrier()
        t0 = time.perf_counter()
        r = dropin.reducer(op, its, n_set0=n_tracks // 2)
        runs, _ = dropin.drain_blocks(r)
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=ctx.cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hs.free(); hf.free(); hv.free()
        return {"bp_per_s": ctx.world * L / float(t.item()), "seconds": float(t.item()), "bp_per_rank": L, "ranks": ctx.world,
                "note": "bulk leg of e2e per rank, one chromosome each, every GPU on its own PCIe link; whole-node bp / max over ranks"}
    except Exception as e:
        return {"error": repr(e)[:300]}


# ---------------------------------------------------------------------------------------------
class Ctx:
    """What every measurement needs: device, distributed state."""
    def __init__(self, device, rank, world, store, shard):
        self.device, self.rank, self.world, self.store, self.shard = device, rank, world, store, shard
        self.replicas = world > 1 and shard == "replicas"
        self.cdev = device
        self.use_dist = world > 1       # --force-dist: the collectives also run with a world of ONE rank (RCCL initialised on a 1-GPU box)


def measure(ctx, name, ops, N, chrom_ids, mean_run, steps, warmup, scale=1.0, n_set0_arg=-1, f64=False, want_moments=False, values="k8"):
    """`steps` timed passes of `ops` over the chromosomes `chrom_ids` (GRCh38 lengths x scale), N tracks of mean run
    `mean_run`: the tracks of one chromosome at a time are generated in HBM (untimed), then window index + fused
    multiplex / reduce kernels run with inputs and outputs resident (timed).  Returns the record of this
    configuration on rank 0 (None elsewhere): value, ms_per_step, roofline, bookkeeping.
    f64: the tracks are handed over as float64 values (what an upstream operator produces): the general kernel."""
    import torch
    import torch.distributed as dist
    from wiggletools_amd import engine, synthgen
    device, world, rank, store, replicas = ctx.device, ctx.world, ctx.rank, ctx.store, ctx.replicas
    use_dist = ctx.use_dist
    chrom_lens = {c: max(int(GRCH38[c] * scale), 1) for c in chrom_ids}
    queue = sorted(chrom_ids, key=lambda c: -chrom_lens[c])         # host-side work queue: largest first
    genome_bp = sum(chrom_lens.values())
    two = any(engine.opcode(o) in (10, 11) for o in ops)
    n_set0 = (n_set0_arg if n_set0_arg >= 0 else N // 2) if two else 0
    stream = torch.cuda.current_stream().cuda_stream
    seed = SEED + (rank if replicas else 0)

    def my_items(pass_no):
        """Work queue of one pass.  One GPU / replicas: every chromosome.  Sharded genome: tickets
        from a counter in the rendezvous store (dynamic), else a static largest-first deal."""
        if (world == 1 and not (use_dist and store is not None)) or replicas:
            yield from queue
        elif store is not None:
            while True:
                k = store.add("wt_queue_%s_%d" % (name, pass_no), 1) - 1
                if k >= len(queue):
                    return
                yield queue[k]
        else:
            load = [0] * world
            for c in queue:
                r = int(np.argmin(load))
                load[r] += chrom_lens[c]
                if r == rank:
                    yield c

    agg = dict(bp=0.0, auc=0.0, runs=0.0, intervals=0.0, windows=0.0)
    moments, stats_last, per_item = {}, {}, {}
    gen_s = [0.0]

    def one_pass(pass_no, record):
        hot_s = 0.0
        idx_ms = red_ms = red_call_ms = 0.0
        for c in my_items(pass_no):
            t0 = time.perf_counter()
            seg, s, f, v = synthgen.device_tracks(seed, [chrom_lens[c]], N, mean_run, 0.02, 800, device, chrom_ids=[c])
            if values in ("full", "fullm"):
                # every mantissa bit in use, and one value in a million 2^-60 times too small for its window to be summed
                # exactly: those windows go to the patch kernel (the generator's k/8 values are the friendliest input the
                # exactness proof can get; this is what less friendly data costs)
                g = torch.arange(v.numel(), device=v.device, dtype=torch.int64)
                h = (((g * 2654435761) ^ (g >> 7)) & 0x7FFFFF).to(torch.float32)
                v = (v + 0.125) * (1.0 + h * (2.0 ** -23))
                if values == "full":
                    v[::1000003] *= 2.0 ** -60
                del g, h
            if f64:
                v = v.double()
            ts = engine.TrackSet.from_device(1, N, seg, s, f, v, np.zeros(N))
            out = ts.alloc_runs()
            torch.cuda.synchronize()
            gen_s[0] += time.perf_counter() - t0
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(1 + 2 * len(ops))]
            # ---- timed: inputs resident, outputs resident ----
            t0 = time.perf_counter()
            ev[0].record()
            for j, op in enumerate(ops):
                if j == 0:
                    ts.index(op, stream)        # reducers of one item share the window index (same window width)
                ev[1 + 2 * j].record()
                ts.reduce(op, out, n_set0=n_set0, stream=stream, sync=False)
                ev[2 + 2 * j].record()
                # THE KERNEL'S duration: the library's own HIP events, recorded on the launch stream immediately before and after the
                # launch (csrc/wt_engine.hip ev_r0 / ev_r1: behind the memsets of the look-back words, around the kernel and -- when
                # windows are patched -- its patch kernels); reading them waits for this launch.  The events of THIS file around the
                # whole call also see the three memsets and the host's way from the record to the launch (Python, ctypes, plan, four
                # API calls) while the stream stands still: 0.1-0.3 ms per launch, 5-10 % of C2's kernels (round 6:
                # tools/experiments/r6_launch_gap.sh -- rocprofv3's kernel durations agree with the library's events).
                st_j = ts.stats()
                red_ms += st_j["reduce_ms"]
                if j == 0:
                    idx_ms += st_j["index_ms"]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # ---- untimed bookkeeping ----
            hot_s += dt
            for j in range(len(ops)):
                red_call_ms += ev[1 + 2 * j].elapsed_time(ev[2 + 2 * j])
            if record:
                n_runs = ts.reduce(ops[-1], out, n_set0=n_set0, stream=stream, sync=True)
                st = ts.stats()
                agg["bp"] += st["covered_bp"]; agg["runs"] += n_runs; agg["intervals"] += int(seg[-1])
                agg["windows"] += st["n_windows"] * len(ops)
                agg["auc"] += out.auc()
                stats_last.update(st)
                agg["patched"] = agg.get("patched", 0) + st.get("patched_windows", 0)
                per_item[c] = dt * 1e3
                if want_moments or use_dist:
                    # Pearson moments of tracks 0 and 1 of this chromosome (scalar gather readiness)
                    a, b = int(seg[0]), int(seg[2])
                    ts2 = engine.TrackSet.from_device(1, 2, seg[:3] - seg[0], s[a:b], f[a:b], v[a:b].float() if f64 else v[a:b], np.zeros(2))
                    moments[c] = ts2.pearson_moments()
                    ts2.close()
            ts.close()
            del ts, out, s, f, v
        return hot_s, idx_ms, red_ms, red_call_ms

    for w in range(max(warmup, 0)):
        one_pass(-1 - w, False)

    pass_s, idx_tot, red_tot, red_call_tot = [], 0.0, 0.0, 0.0
    for k in range(steps):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        hs, im, rm, rcm = one_pass(k, k == steps - 1)
        t = torch.tensor([hs], dtype=torch.float64, device=ctx.cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)        # a pass is over when the slowest rank is
            dist.barrier()
        pass_s.append(float(t.item()))
        idx_tot += im; red_tot += rm; red_call_tot += rcm
    elapsed = float(sum(pass_s))

    # genome-wide scalars (RCCL over xGMI when world > 1): sums by all_reduce, Pearson by all_gather + ordered merge
    vec = torch.tensor([agg["bp"], agg["auc"], agg["runs"], agg["intervals"], agg["windows"], idx_tot, red_tot, gen_s[0], red_call_tot],
                       dtype=torch.float64, device=ctx.cdev)
    mom = torch.zeros((len(GRCH38), 6), dtype=torch.float64, device=ctx.cdev)
    for c, m in moments.items():
        mom[c] = torch.tensor(m, dtype=torch.float64, device=ctx.cdev)
    queue_check = None
    if use_dist:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        gathered = [torch.zeros_like(mom) for _ in range(world)]
        dist.all_gather(gathered, mom)
        mom = gathered[0] if replicas else torch.stack(gathered).sum(0)   # every row is computed by exactly one rank
        # the work queue of the record pass: who got which chromosome
        mine = torch.zeros(len(GRCH38), dtype=torch.int64, device=ctx.cdev)
        for c in per_item:
            mine[c] = 1
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        cnt = torch.stack(got).sum(0).cpu().numpy()
        want = np.zeros(len(GRCH38), np.int64)
        want[list(chrom_ids)] = 1 if not replicas else world
        queue_check = {"every_chromosome_exactly_once": bool((cnt == want).all()), "chromosomes_per_rank": [int(g.sum().item()) for g in got],
                       "how": "tickets from a counter in the rendezvous store" if store is not None and not replicas else ("replicas" if replicas else "static deal")}
    tot_bp, tot_auc, tot_runs, tot_int, tot_win, idx_all, red_all, gen_all, red_call_all = [float(x) for x in vec.tolist()]
    pearson = None
    if moments or use_dist:
        from wiggletools_amd import shard
        rows = mom.cpu().numpy()
        order = sorted(chrom_ids, key=lambda c: ("chr%d" % (c + 1)).encode())       # strcmp order (multiplexer.c:56)
        pearson = shard.pearson_from_moments([rows[c] for c in order if rows[c][0] > 0])
    if rank != 0:
        return None

    passes = steps
    bp_per_pass = tot_bp                                # the record pass: every rank's share, summed
    value = bp_per_pass * passes / elapsed
    kern = stats_last.get("kernel", 0)
    vt = "f64" if f64 else "f32"
    kernel = ("wt_delta_kernel<%s>" % ops[-1]) if kern == 1 else ("wt_walk_kernel" if kern == 2 else ("wt_mwalk_kernel" if kern == 3 else "wt_reduce_kernel<%s,%s>" % (ops[-1], vt)))
    # algorithmic bytes of the fused multiplex+reduce launches of ONE pass: every input run read
    # once per launch (start, finish, value = 12 B; 16 B for float64 values), the two window-index rows per
    # window, every output run written once (start, finish, f64 value = 16 B).  DESIGN.md 4.6.
    # Launches differ in size (one per chromosome), so the rate is bytes of all launches of a pass /
    # their summed durations (HIP events on the launch stream, around every launch) -- the
    # duration-weighted mean of the per-launch rates; with N GPUs: the per-GPU mean.
    alg_bytes = len(ops) * ((16.0 if f64 else 12.0) * tot_int + 16.0 * tot_runs) + 8.0 * N * tot_win
    kernel_ms_sum = red_all / passes                    # per pass, summed over launches (and ranks)
    achieved = alg_bytes / (kernel_ms_sum * 1e-3) / 1e9
    tile_equiv = len(ops) * tot_runs * (4.0 * N + N / 8.0 + 24.0) / (kernel_ms_sum * 1e-3) / 1e9
    # measured HBM traffic of THIS kernel: profiles/traffic.json keeps, per kernel label, the ratio of a PMC pass's FETCH_SIZE + WRITE_SIZE
    # to that launch's algorithmic bytes (tools/round6.sh); a kernel without a pass of its own gets null -- never another kernel's ratio
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = (tj.get("kernels") or {}).get(kernel)
            if ent is None and tj.get("kernel") == kernel:
                ent = tj                                   # (round 5's single-kernel file)
            if ent is not None:
                traffic = ent["hbm_bytes_per_algorithmic_byte"] * alg_bytes
                traffic_src = "PMC passes of %s (profiles/traffic.json: FETCH_SIZE + WRITE_SIZE per launch, ratio to that launch's algorithmic bytes) scaled to this launch" % ent.get("profile", "an earlier profile")
        except Exception:
            traffic = None
    bound = "hbm"
    note = None
    issue = None
    if kern in (0, 2, 3):
        o = ops[-1]
        if o in ("median", "wilcoxon", "mwu"):
            bound, note = "issue", (("median by walking (csrc/wt_walk.h)" if kern == 2 else ("MWU by walking (csrc/wt_mwalk.h)" if kern == 3 else "register-column reducer")) +
                                    ": bound by instruction issue and latency, not by HBM -- the frac against the HBM peak is "
                                    "reported for the record only (DESIGN A.1); roofline.issue prices it against the VALU issue peak")
            # VALU instructions per output run from the SQ counters of an earlier profile (profiles/issue.json, like
            # traffic.json for the HBM bytes) x this run's output runs / this run's kernel time, against what the chip
            # can issue: 256 CUs x 4 SIMDs x one wave-wide VALU instruction per 4 cycles at 2.4 GHz
            ipath = os.path.join(ROOT, "profiles", "issue.json")
            if os.path.exists(ipath):
                try:
                    ij = json.load(open(ipath)).get(("median_walk" if kern == 2 else "median") if o == "median" else ("wilcoxon_walk" if kern == 3 else "wilcoxon"))
                    if ij:
                        peak = 256 * 4 * 2.4e9 / 4.0
                        ach = ij["valu_instructions_per_output_run"] * tot_runs / (kernel_ms_sum * 1e-3)
                        issue = {"bound": "valu issue", "achieved": ach, "peak": peak, "unit": "wave-wide VALU instructions/s", "frac": ach / peak,
                                 "valu_instructions_per_output_run": ij["valu_instructions_per_output_run"],
                                 "salu_instructions_per_output_run": ij.get("salu_instructions_per_output_run"),
                                 "source": ij.get("source")}
                except Exception:
                    issue = None
        else:
            bound, note = "valu", "f32->f64 widen + add per (track, position): VALU bound (profiles/: VALUBusy)"
    return {
        "metric": "genomic bp/s (whole node) for 'mean' over N BigWig tracks" if ops == ["mean"] else
                  "genomic bp/s (whole node) for '%s' over N tracks" % "+".join(ops),
        "value": value, "unit": "genomic bp/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / passes * 1e3,
        "higher_is_better": True, "scaling": "weak" if replicas else "strong", "vs_baseline": None,
        # what `value` is: index + fused multiplex / reduce kernels over run lists RESIDENT in HBM (SURVEY 8d metric 2's
        # material, the contract's "inputs already resident"); files -> result on the host is config.north_star_files_to_result
        "value_kind": "resident kernels (window index + fused multiplex/reduce), inputs and outputs in HBM",
        # the arithmetic type of the path: the reference accumulates in f64; the difference-array kernel does it in
        # exact int64 / 128-bit integers of the scaled float32 mantissas, which IS that f64 result
        "dtype": "f64", "accumulation": "exact int64 / 128-bit integer sums, rounded once to f64" if kern == 1 else "f64", "data": "synthetic",
        # (the driver's record truncates strings at 120 characters: the workload fits, the prose is in workload_note)
        "config": {"workload": "%s: %s, %d %s tracks, %d chrom(s) GRCh38 x %g = %.3f Gbp/step, run %g bp, resident in HBM"
                               % (name, "+".join(ops), N, "f64" if f64 else "f32", len(chrom_ids), scale, genome_bp / 1e9, mean_run),
                   "workload_note": "synthetic run-list tracks, 2% gaps; one step = one whole pass, chromosomes generated in HBM one at a time (untimed) and processed resident (timed)",
                   "config": name, "ops": ops, "tracks": N, "mean_run_bp": mean_run,
                   "genome_bp": genome_bp, "covered_bp_per_step": bp_per_pass,
                   "input_runs_per_step": tot_int, "output_runs_per_step": tot_runs,
                   "window_bp": stats_last.get("window_bp"), "lds_bytes_per_workgroup": stats_last.get("lds_bytes"),
                   "values": {"k8": "k/8, k < 800 (exact in f32)", "fullm": "full mantissas in [0.125, 200) (ordinary signal: no outliers)",
                              "full": "full mantissas in [0.125, 200), one in 1 000 003 scaled by 2^-60"}[values],
                   "windows_per_step": tot_win, "patched_windows_per_step": agg.get("patched", 0),      # (of the record pass; rank 0's share of the patched ones)
                   "sharding": ("replicas: every rank walks its own genome" if replicas else
                                "one genome, chromosomes from a shared host-side work queue (store counter), no data-path collective")
                               if world > 1 else "single GPU"},
        "roofline": {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes, "launch": "the %d launches of one pass (one per chromosome%s)" % (len(per_item) * len(ops) if world == 1 else int(len(chrom_ids) * len(ops)), ", per reducer" if len(ops) > 1 else ""),
                     "kernel_ms": kernel_ms_sum, "index_kernel_ms": idx_all / passes,
                     "reduce_call_ms": red_call_all / passes,     # stream events around the whole reduce call: + memsets + the host's launch path
                     "timing": "HIP events on the launch stream immediately around every launch (the library's, read here); reduce_call_ms: events around the whole call",
                     "frac_with_index": alg_bytes / ((kernel_ms_sum + idx_all / passes) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "tile_equivalent_GBs": tile_equiv, "note": note, "issue": issue},
        "auc_check": tot_auc, "pearson_tracks_0_1": pearson, "output_runs": tot_runs, "work_queue_check": queue_check,
        "collectives": ({"backend": dist.get_backend(), "world": world, "ran": ["barrier", "all_reduce(max) of the pass time", "all_reduce(sum) of the scalars",
                                                                                "all_gather of the Pearson moments", "all_gather of the work-queue record"]} if use_dist else None),
        "gen_seconds_total": gen_all, "pass_seconds": pass_s,
        "_chrom_lens": [chrom_lens[c] for c in chrom_ids],
    }


def with_cpu(res, chrom_ids, ops, N, mean_run, many_core, target_s=12.0):
    lens = res.pop("_chrom_lens")
    cb = cpu_baseline(chrom_ids, ops[-1], N, mean_run, lens, many_core=many_core, target_s=target_s)
    res["cpu_baseline"] = cb
    res["speedup_vs_cpu_baseline"] = res["value"] / cb["value"]
    if "many_core" in cb and "value" in cb["many_core"]:
        res["speedup_vs_many_core_cpu"] = res["value"] / cb["many_core"]["value"]
    return res


def slim(res):
    """A sub-record of the driver line: the figures, without the long prose of the main record."""
    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline", "cpu_baseline", "speedup_vs_cpu_baseline",
            "speedup_vs_many_core_cpu", "auc_check", "output_runs", "pearson_tracks_0_1")
    out = {k: res[k] for k in keep if k in res}
    out["workload"] = res["config"]["workload"]
    out["patched_windows_per_step"] = res["config"].get("patched_windows_per_step")
    out["windows_per_step"] = res["config"].get("windows_per_step")
    for k in ("traffic_source", "launch", "note"):
        out["roofline"].pop(k, None)
    if "cpu_baseline" in out:
        for k in ("host_cores_note",):
            out["cpu_baseline"].pop(k, None)
    return out


COMPACT_LINE_LIMIT = 6000      # bytes: round 5's 21.5 KB line was not parsed by the driver (round 4's 19.3 KB one was)


def _sig(v, digits=5):
    """Floats of the compact line at `digits` significant digits (the full record keeps every bit)."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def compact(res):
    """The driver's line: the contract's keys, scalar `config` keys, `roofline` and `cpu_baseline` without prose.

    Everything else (the nested sub-records of the other configurations, the e2e legs, pass times) goes to the
    full record (`emit`).  Kept under COMPACT_LINE_LIMIT bytes: tests/test_bench_line.py holds it there."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "accumulation", "data", "value_kind")
    out = {k: res.get(k) for k in top if k in res}
    cfg = {}
    for k, v in (res.get("config") or {}).items():
        if isinstance(v, str):
            if k in ("workload", "config", "sharding", "values"):
                cfg[k] = v[:120]
        elif isinstance(v, (int, float)) or v is None:
            cfg[k] = v
        elif k == "ops":
            cfg[k] = v
    out["config"] = cfg
    rf = dict(res.get("roofline") or {})
    for k in ("traffic_source", "launch", "note", "issue"):
        rf.pop(k, None)
    out["roofline"] = rf
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "host_cores") if k in cb}
        c["sample"] = str(cb.get("sample", ""))[:160]
        mc = cb.get("many_core")
        if isinstance(mc, dict) and "value" in mc:
            c["many_core_value"] = mc["value"]
            c["many_core_cores"] = mc.get("cores")
        out["cpu_baseline"] = c
    for k in ("speedup_vs_cpu_baseline", "speedup_vs_many_core_cpu", "output_runs", "auc_check", "pearson_tracks_0_1", "bench_seconds", "full_record"):
        if k in res:
            out[k] = res[k]
    # multi-GPU runs: which collectives ran on which backend, and the work queue's check (small dicts)
    for k in ("collectives", "work_queue_check"):
        if isinstance(res.get(k), dict) and len(json.dumps(res[k])) < 700:
            out[k] = res[k]
    out = _sig(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LINE_LIMIT:          # never lose the line to its own extras: drop config's tail first
        keys = list(out["config"])
        while len(line) > COMPACT_LINE_LIMIT and len(keys) > 1:
            out["config"].pop(keys.pop())
            line = json.dumps(out, separators=(",", ":"))
    return line


def emit(res, full_path):
    """Full record -> a file (and stderr); compact record -> the LAST line of stdout (what the driver parses)."""
    full = json.dumps(res)
    if full_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)) or ".", exist_ok=True)
            with open(full_path, "w") as f:
                f.write(full + "\n")
            res["full_record"] = full_path
        except OSError as e:
            res["full_record"] = "not written: %r" % (e,)
    sys.stderr.write("bench full record: " + full + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-record", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "bench_full_last.json"),
                    help="where the full record (every nested sub-record) is written; the last stdout line is the compact one")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--op", default=None, help="override the configuration's reducer(s), comma separated")
    ap.add_argument("--tracks", type=int, default=None)
    ap.add_argument("--mean-run", type=float, default=16.0)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the GRCh38 chromosome lengths (1 = 3.1 Gbp)")
    ap.add_argument("--n-set0", type=int, default=-1, help="two-sample ops: tracks in the first set (default N/2)")
    ap.add_argument("--shard", default="genome", choices=["genome", "replicas"])
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, collectives on the host) -- a does-it-run check, not a scaling number")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (and with backend nccl: RCCL) even for a world of one rank, and run the work queue + scalar collectives through it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-many-core", action="store_true")
    ap.add_argument("--chroms", default=None, help="experiments: only these chromosomes of the configuration (comma separated indices)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--moments", action="store_true", help="also compute the Pearson moments of tracks 0, 1 per chromosome (always on with N > 1)")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub-records (c3 / c4 / c5, mean run 1 / 200, other kernels) of the default line")
    ap.add_argument("--sub-steps", type=int, default=2, help="timed passes of every sub-record")
    ap.add_argument("--f64", action="store_true", help="hand the tracks over as float64 values (general kernel)")
    ap.add_argument("--values", default="k8", choices=["k8", "fullm", "full"], help="k8: the generator's k/8 values; full: full mantissas and one value in a million outside its window's exact range (patched windows)")
    ap.add_argument("--e2e-bw-mbp", type=float, default=248.956422, help="chromosome length of the BigWig-files-to-result leg (0: skip); default chromosome 1")
    ap.add_argument("--e2e-mbp", type=float, default=248.956422, help="chromosome length of the end-to-end (drop-in layer) leg; default chromosome 1")
    ap.add_argument("--no-genome-files", action="store_true", help="skip the whole-genome BigWig-files-to-result leg (e2e_bigwig_genome)")
    ap.add_argument("--chr1-files", action="store_true", help="also run round 3's one-chromosome file leg (e2e_bigwig: chromosome 1 x N files)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    store = None
    if world > 1 or args.force_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:     # a free port (a fixed one collides between concurrent runs)
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
    ctx = Ctx(device, rank, world, store, args.shard)
    ctx.cdev = torch.device("cpu") if args.dist_backend == "gloo" else device       # where collectives' tensors live
    ctx.use_dist = world > 1 or args.force_dist

    cfg = CONFIGS[args.config]
    ops = args.op.split(",") if args.op else cfg["ops"]
    N = args.tracks if args.tracks else cfg["tracks"]
    chrom_ids = cfg["chroms"] if not args.chroms else [int(x) for x in args.chroms.split(",")]
    t_start = time.perf_counter()
    res = measure(ctx, args.config, ops, N, chrom_ids, args.mean_run, args.steps, args.warmup, scale=args.scale,
                  n_set0_arg=args.n_set0, f64=args.f64, want_moments=args.config == "c5" or args.moments, values=args.values)
    default_line = args.config == "c2" and not args.op and not args.tracks and not args.chroms and args.scale == 1.0 and not args.f64 and args.values == "k8"
    # multi-GPU: the e2e bulk leg per rank (every GPU has its own PCIe link) -- all ranks take part
    e2e_multi = None
    if world > 1 and not args.no_e2e:
        e2e_multi = e2e_sharded(ctx, ops[-1], N, args.mean_run, min(args.e2e_mbp, 100.0))
        trim_pools()
    if rank == 0:
        if world == 1 and not args.no_e2e:
            fit = min(1.0, 100.0 / N) * (args.scale if args.scale < 1 else 1.0)
            try:
                res["e2e"] = e2e_dropin(ops[-1], N, args.mean_run, args.e2e_mbp * fit, device)
            except Exception as e:      # never lose the bench line to an extra leg
                res["e2e"] = {"error": repr(e)[:300]}
            if args.e2e_bw_mbp > 0 and args.chr1_files:
                try:
                    res["e2e_bigwig"] = e2e_bigwig(ops[-1], N, args.mean_run, args.e2e_bw_mbp * fit, device)
                except Exception as e:
                    res["e2e_bigwig"] = {"error": repr(e)[:300]}
            trim_pools()
            if not args.no_genome_files:
                # THE north-star figure (BASELINE.json): `mean` over 100 whole-genome BigWig files -> result, >= 1e9 bp/s
                try:
                    gscale = genome_file_scale(N, args.mean_run) * (args.scale if args.scale < 1 else 1.0)
                    g = e2e_bigwig_genome(ops[-1], N, args.mean_run, gscale, device, fresh_process=True) if gscale > 0.001 else {"error": "no room for the files"}
                except Exception as e:
                    g = {"error": repr(e)[:300]}
                res["e2e_bigwig_genome"] = g
                trim_pools()
                # SCALAR keys of `config`: the part of the line the driver keeps (round 4 nested them in a dict, which the
                # driver's record dropped).  cold = what a CLI process gets: the process's first pipe, nothing pooled.
                cfgd = res["config"]
                cfgd["e2e_files_what"] = "%s, %d whole-genome BigWig files (24 chroms, %.3f Gbp), open -> last run on host" % (ops[-1], N, g.get("bp", 0) / 1e9)
                cfgd["e2e_files_cold_bp_per_s"] = g.get("bp_per_s")
                cfgd["e2e_files_warm_bp_per_s"] = g.get("warm_bp_per_s")
                cfgd["e2e_files_steady_bp_per_s"] = g.get("steady_bp_per_s")
                cfgd["e2e_files_cold_seconds"] = (g.get("cold") or {}).get("seconds")
                cfgd["e2e_files_target_bp_per_s"] = 1e9
                fp = g.get("fresh_process") or {}
                cfgd["e2e_files_fresh_process_bp_per_s"] = fp.get("bp_per_s")                   # first library call -> last run, new process
                cfgd["e2e_files_fresh_process_seconds"] = fp.get("seconds")
                cfgd["e2e_files_fresh_process_library_load_s"] = fp.get("library_load_s")       # dlopen: the HIP runtime's shared objects
                # ... and the same before anything had read the freshly written files (the page cache's first read of 90 GB included)
                cfgd["e2e_files_first_touch_bp_per_s"] = (g.get("fresh_process_first_touch") or {}).get("bp_per_s")
                if g.get("error"):
                    cfgd["e2e_files_error"] = str(g.get("error"))[:110]
                # the same pipeline on files written at zlib level 6 (libBigWig's / wigToBigWig's default; SURVEY 8d's stored
                # form is level 1) next to level 1 on the same two chromosomes (21, 22: writing 90 GB at level 6 would take
                # the bench ten minutes of host zlib)
                if (gscale >= 0.999 or os.environ.get("WTAMD_BENCH_FORCE_LEVELS")) and not os.environ.get("WTAMD_BENCH_NO_LEVELS"):
                    lv = {}
                    for level in (1, 6):
                        try:
                            q = e2e_bigwig_genome(ops[-1], N, args.mean_run, 1.0, device, only=[20, 21], level=level)
                            lv["z%d" % level] = {k: q.get(k) for k in ("zlib_level", "bp", "file_bytes", "file_bytes_per_bp", "files_written_s", "sections",
                                                                       "bp_per_s", "warm_bp_per_s", "steady_bp_per_s")}
                            lv["z%d" % level]["warm_sum_device_decode_ms"] = (q.get("warm") or {}).get("sum_device_decode_ms")
                            lv["z%d" % level]["warm_seconds"] = (q.get("warm") or {}).get("seconds")
                        except Exception as e:
                            lv["z%d" % level] = {"error": repr(e)[:300]}
                        trim_pools()
                    res["e2e_bigwig_levels"] = lv
                    # (97 Mbp: a quarter of a second, half of it start-up -- the steady figures compare the decoders)
                    cfgd["e2e_files_z6_steady_bp_per_s_chr21_22"] = lv.get("z6", {}).get("steady_bp_per_s")
                    cfgd["e2e_files_z1_steady_bp_per_s_chr21_22"] = lv.get("z1", {}).get("steady_bp_per_s")
                    cfgd["e2e_files_z6_warm_bp_per_s_chr21_22"] = lv.get("z6", {}).get("warm_bp_per_s")
                    cfgd["e2e_files_z1_warm_bp_per_s_chr21_22"] = lv.get("z1", {}).get("warm_bp_per_s")
                res["value_e2e_bigwig_genome"] = g.get("bp_per_s")
                res["value_e2e_bigwig_genome_warm"] = g.get("warm_bp_per_s")
                res["value_e2e_bigwig_genome_steady"] = g.get("steady_bp_per_s")
            # SURVEY 8d metric (1), first pop -> last result on the host, next to the resident-kernel `value`
            res["value_e2e_bulk"] = (res["e2e"].get("bulk") or {}).get("bp_per_s")
            res["value_e2e_bigwig"] = res.get("e2e_bigwig", {}).get("bp_per_s")
            res["value_e2e_bigwig_warm"] = res.get("e2e_bigwig", {}).get("warm_bp_per_s")
            res["value_e2e_bigwig_steady"] = res.get("e2e_bigwig", {}).get("steady_bp_per_s")
        if e2e_multi is not None:
            res["e2e_sharded"] = e2e_multi
        if world == 1 and not args.no_cpu_baseline:
            with_cpu(res, chrom_ids, ops, N, args.mean_run, not args.no_many_core)
        res.pop("_chrom_lens", None)
    # the other BASELINE configurations and run lengths, as sub-records of the default line (one GPU)
    if world == 1 and default_line and not args.no_sub:
        subs, runs, others = {}, {}, {}

        def sub(name, ops_, N_, chroms_, mean_run_, f64=False, cpu=True, moments=False, values="k8"):
            try:
                r = measure(ctx, name, ops_, N_, chroms_, mean_run_, args.sub_steps, 2, f64=f64, want_moments=moments, values=values)
                if cpu and not args.no_cpu_baseline:
                    # (one evaluation thread, a 6 s sample: the many-core figure of these configurations is in DESIGN.md 5
                    #  from `bench.py --config c3 / c4 / c5`, which still measures it; the default line spends its minutes
                    #  on the whole-genome file set)
                    with_cpu(r, chroms_, ops_, N_, mean_run_, False, target_s=6.0)
                r.pop("_chrom_lens", None)
                return slim(r)
            except Exception as e:
                return {"error": repr(e)[:300]}

        for name in ("c3", "c4", "c5"):
            c = CONFIGS[name]
            subs[name] = sub(name, c["ops"], c["tracks"], c["chroms"], args.mean_run, moments=name == "c5")
        # the other two-sample reducer north_star names (Welch t-test, setComparisons.c:35-121), same shape as C5
        subs["ttest"] = sub("ttest", ["ttest"], 100, list(range(24)), args.mean_run)
        # C2 at the other run lengths of SURVEY 8d: mean run 1 bp on chromosomes 19-22 + Y (their 100 dense tracks fit HBM
        # one chromosome at a time: 77 GB for chromosome 19), mean run 200 bp on the whole genome
        runs["l1"] = sub("c2/l=1", ["mean"], 100, [18, 19, 20, 21, 23], 1.0, cpu=False)
        runs["l200"] = sub("c2/l=200", ["mean"], 100, list(range(24)), 200.0, cpu=False)
        # C2's kernel on data the exactness proof likes less: full mantissas, and one value in a million far outside its
        # window's exact range (those windows are redone by the patch kernel), chromosome 21
        runs["full_mantissa"] = sub("c2/full mantissas", ["mean"], 100, [20], args.mean_run, cpu=False, values="fullm")
        runs["full_mantissa_patched"] = sub("c2/full mantissas + patched windows", ["mean"], 100, [20], args.mean_run, cpu=False, values="full")
        # kernels the headline does not exercise, N = 100 on chromosome 21 (46.7 Mbp)
        others["max"] = sub("max", ["max"], 100, [20], args.mean_run, cpu=False)
        others["product"] = sub("product", ["product"], 100, [20], args.mean_run, cpu=False)
        others["mean_f64_values"] = sub("mean/f64", ["mean"], 100, [20], args.mean_run, f64=True, cpu=False)
        others["sum_500"] = sub("sum/500", ["sum"], 500, [20], args.mean_run, cpu=False)
        # C5's reducer on values without ties (real float signal rarely ties; the generator's 800 levels tie at nearly every position):
        # the library samples the values and takes the walking kernel here, the register columns there (csrc/wt_engine.hip wt_mwu_few_ties)
        others["wilcoxon_full_mantissa"] = sub("wilcoxon/full mantissas", ["wilcoxon"], 100, [20], args.mean_run, cpu=False, values="fullm")
        if rank == 0:
            res["configs"] = subs
            res["c2_runs"] = runs
            res["other_kernels"] = others
            # ... and their headline figures as scalar keys of `config` (what the driver's record keeps)
            def put(key, rec, what):
                if isinstance(rec, dict) and "error" not in rec:
                    v = rec.get("roofline", {}).get(what) if what in ("frac", "kernel_ms") else rec.get(what)
                    if v is not None:
                        res["config"][key] = v
            for name in ("c3", "c4", "c5", "ttest"):
                put("%s_hbm_frac" % name, subs.get(name), "frac")
                put("%s_bp_per_s" % name, subs.get(name), "value")
                put("%s_ms_per_step" % name, subs.get(name), "ms_per_step")
            put("c2_l1_hbm_frac", runs.get("l1"), "frac")
            put("c2_l200_hbm_frac", runs.get("l200"), "frac")
            put("c2_full_mantissa_hbm_frac", runs.get("full_mantissa"), "frac")
            put("c2_patched_hbm_frac", runs.get("full_mantissa_patched"), "frac")
            for name in ("max", "product", "mean_f64_values", "sum_500", "wilcoxon_full_mantissa"):
                put("%s_hbm_frac" % name, others.get(name), "frac")
    if rank == 0:
        res["bench_seconds"] = time.perf_counter() - t_start
        emit(res, args.full_record)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
