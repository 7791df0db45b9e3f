#!/usr/bin/env python
"""bench.py -- throughput of the hot path (Multiplexer -> MeanReduction) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`;
for N>1 launched by torch.distributed.run, one rank per GPU.  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1]): `mean` over 100 synthetic BigWig-like
tracks (float32 values k/8, run length ~ Geometric(1/l), 2 % gaps),
laid out on 24 chromosomes with GRCh38 proportions.  One STEP = one full pass of
the hot path over the resident batch: window-index kernel + fused
multiplex/reduce kernel, inputs and outputs in HBM.  The batch is the genome
scaled by --scale (default 1/8 at l=16 so that generation + K steps finish in
minutes); bp/s is intensive, a whole genome is 1/scale steps.

Multi-GPU: chromosome batches are independent: every rank owns its
own batch (weak scaling), no collective on the data path; one RCCL all_reduce of
the genome-wide AUC / bp scalars after the timed region.
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


def synth_device(n_tracks, chrom_lens, mean_run, gap_prob, seed, device):
    """Synthetic run lists generated on the GPU with torch (plumbing only)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    S, F, V = [], [], []
    seg_off = [0]
    p = 1.0 / mean_run
    for clen in chrom_lens:
        for _ in range(n_tracks):
            n_est = int(clen / mean_run * 1.1) + 4096
            # (float32 geometric_ occasionally yields 0: a zero-length run is undefined behaviour for
            #  the reference's Multiplexer, SURVEY appendix A, and for wtamd_tracks -- hence the clamp)
            lens = torch.empty(n_est, device=device, dtype=torch.float32).geometric_(p, generator=g).to(torch.int64).clamp_(min=1) \
                if mean_run > 1 else torch.ones(n_est, device=device, dtype=torch.int64)
            ends = torch.cumsum(lens, 0)
            k = int(torch.searchsorted(ends, torch.tensor([clen], device=device, dtype=torch.int64)).item()) + 1
            if k > n_est:      # extremely unlikely: estimate too small
                k = n_est
            ends = ends[:k].clone()
            ends[-1] = clen
            starts = torch.cat([torch.zeros(1, device=device, dtype=torch.int64), ends[:-1]])
            keep = torch.rand(k, device=device, generator=g) >= gap_prob
            vals = torch.randint(0, 800, (k,), device=device, generator=g).to(torch.float32) / 8.0
            S.append((starts[keep] + 1).to(torch.int32))
            F.append((ends[keep] + 1).to(torch.int32))
            V.append(vals[keep])
            seg_off.append(seg_off[-1] + int(S[-1].numel()))
            del lens, ends, starts, keep, vals
    start = torch.cat(S); del S
    finish = torch.cat(F); del F
    value = torch.cat(V); del V
    return np.array(seg_off, np.int64), start, finish, value


def cpu_baseline(seg_off, start, finish, value, n_chrom, n_tracks, target_s=15.0):
    """Times the COMPILED REFERENCE (oracle/_ref, else our C restatement) on a bounded
    sample: the first `sample_bp` positions of chromosome 0 of this very batch."""
    from oracle import oracle as O
    O.build()
    have_ref = O.have_ref()

    def sample(sample_bp):
        so = [0]
        S, F, V = [], [], []
        for i in range(n_tracks):
            lo, hi = int(seg_off[i]), int(seg_off[i + 1])
            s = start[lo:hi]
            cut = int((s <= sample_bp).sum().item())
            S.append(start[lo:lo + cut].cpu().numpy())
            f = finish[lo:lo + cut].cpu().numpy().copy()
            np.minimum(f, sample_bp + 1, out=f)
            F.append(f)
            V.append(value[lo:lo + cut].cpu().numpy().astype(np.float64))
            so.append(so[-1] + cut)
        return dict(n_chrom=1, n_tracks=n_tracks, seg_off=np.array(so, np.int64),
                    start=np.concatenate(S), finish=np.concatenate(F), value=np.concatenate(V),
                    defaults=np.zeros(n_tracks))

    def run(d):
        if have_ref:
            sec, runs, bp = O.ref_time_reduce(d, "mean")
            return sec, bp
        t0 = time.perf_counter()
        c, s, f, v = O.reduce(d, "mean")
        return time.perf_counter() - t0, int((f - s).sum())

    probe_bp = 200_000
    sec, bp = run(sample(probe_bp))
    rate = bp / max(sec, 1e-9)
    sample_bp = int(min(max(rate * target_s, probe_bp), 64_000_000))
    sec, bp = run(sample(sample_bp))
    return {"value": bp / sec, "unit": "genomic bp/s", "cores": 1,
            "kind": "reference" if have_ref else "port",
            "sample": "mean over the same %d tracks, first %d bp of chromosome 0 of the bench batch, "
                      "one evaluation thread (the reference never parallelises evaluation), sink=none; %.1f s"
                      % (n_tracks, sample_bp, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tracks", type=int, default=100)
    ap.add_argument("--mean-run", type=float, default=16.0)
    ap.add_argument("--scale", type=float, default=0.125, help="fraction of the GRCh38 lengths per step")
    ap.add_argument("--op", default="mean")
    ap.add_argument("--n-set0", type=int, default=-1, help="two-sample ops: tracks in the first set (default N/2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from wiggletools_amd import engine

    chrom_lens = [max(int(x * args.scale), 1) for x in GRCH38]
    total_bp = sum(chrom_lens)
    t0 = time.perf_counter()
    seg_off, start, finish, value = synth_device(args.tracks, chrom_lens, args.mean_run, 0.02,
                                                 20260927 + rank, device)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    n_intervals = int(seg_off[-1])
    ts = engine.TrackSet.from_device(len(chrom_lens), args.tracks, seg_off, start, finish, value,
                                     np.zeros(args.tracks))
    n_bad, first_bad = ts.validate()
    assert n_bad == 0, "synthetic tracks violate the run-list contract (%d runs, first at %d)" % (n_bad, first_bad)
    out = ts.alloc_runs()
    stream = torch.cuda.current_stream().cuda_stream

    two = engine.opcode(args.op) in (10, 11)
    n_set0 = (args.n_set0 if args.n_set0 >= 0 else args.tracks // 2) if two else 0

    def step(sync=False):
        ts.index(args.op, stream)
        return ts.reduce(args.op, out, n_set0=n_set0, stream=stream, sync=sync)

    for _ in range(max(args.warmup, 0)):
        step(sync=True)
    n_runs = step(sync=True) if args.warmup == 0 else out.n
    st = ts.stats()
    covered_bp = st["covered_bp"]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        ts.index(args.op, stream)
        ev[k][1].record()
        ts.reduce(args.op, out, n_set0=n_set0, stream=stream, sync=False)
        ev[k][2].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    index_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    reduce_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # post-timing verification + scalar gather (RCCL over xGMI when world > 1)
    n_runs = ts.reduce(args.op, out, n_set0=n_set0, stream=stream, sync=True)
    st = ts.stats()
    covered_bp = st["covered_bp"]
    auc = out.auc()
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    agg = torch.tensor([float(covered_bp), auc, float(n_runs), float(n_intervals)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    tot_bp, tot_auc, tot_runs, tot_intervals = [float(x) for x in agg.tolist()]

    if rank == 0:
        N = args.tracks
        # algorithmic bytes of ONE launch of the dominant (fused multiplex+reduce) kernel on this
        # rank: every input run read once (start,finish,value = 12 B), the two window-index rows
        # per window, every output run written once (start,finish,f64 value = 16 B).  DESIGN.md 4.
        alg_bytes = 12.0 * n_intervals + 8.0 * N * st["n_windows"] + 16.0 * n_runs
        achieved = alg_bytes / (reduce_ms * 1e-3) / 1e9
        tile_equiv = n_runs * (4.0 * N + N / 8.0 + 24.0) / (reduce_ms * 1e-3) / 1e9   # SURVEY 8d tile figure
        # HBM bytes per launch from the PMC passes (profiles/traffic.json, measured with
        # tools_prof.sh): stored as a ratio to the algorithmic bytes of the profiled launch and
        # scaled to this launch; None when no profile has been taken for this kernel.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        kernel = ("wt_delta_kernel<%s>" % args.op) if st.get("kernel") == 1 else ("wt_reduce_kernel<%s,f32>" % args.op)
        if os.path.exists(tpath) and args.op == "mean":
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel", "").split("<")[0] == kernel.split("<")[0]:
                    traffic = tj["hbm_bytes_per_algorithmic_byte"] * alg_bytes
            except Exception:
                traffic = None
        res = {
            "metric": "genomic bp/s (whole node) for 'mean' over N BigWig tracks",
            "value": tot_bp * args.steps / elapsed,
            "unit": "genomic bp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if st.get("kernel") == 1 else "f64", "data": "synthetic",
            "config": {"workload": "%s over %d synthetic float32 run-list tracks, 24 chromosomes = GRCh38 x %g "
                                   "(%.0f Mbp per GPU per step), mean run %g bp, 2%% gaps, tracks resident in HBM"
                                   % (args.op, N, args.scale, total_bp / 1e6, args.mean_run),
                       "op": args.op, "tracks": N, "mean_run_bp": args.mean_run, "bp_per_step_per_gpu": total_bp,
                       "input_runs_per_gpu": n_intervals, "output_runs_per_gpu": n_runs,
                       "window_bp": st["window_bp"], "lds_bytes_per_workgroup": st["lds_bytes"],
                       "sharding": "one independent chromosome batch per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": reduce_ms, "index_kernel_ms": index_ms,
                         "tile_equivalent_GBs": tile_equiv},
            "auc_check": tot_auc, "output_runs": tot_runs, "gen_seconds": gen_s,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(seg_off, start, finish, value, len(chrom_lens), N)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
