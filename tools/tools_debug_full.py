"""GPU-box debugging aid: locate runs where the difference-array kernel and the general kernel disagree."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from wiggletools_amd import engine

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
dev = torch.device("cuda", 0)
N = 100
chrom_lens = [max(int(x * scale), 1) for x in bench.GRCH38]
seg_off, start, finish, value = bench.synth_device(N, chrom_lens, 16.0, 0.02, 7, dev)
stream = torch.cuda.current_stream().cuda_stream

def run(op, no_delta):
    if no_delta: os.environ["WTAMD_NO_DELTA"] = "1"
    else: os.environ.pop("WTAMD_NO_DELTA", None)
    ts = engine.TrackSet.from_device(len(chrom_lens), N, seg_off, start, finish, value, np.zeros(N))
    out = ts.alloc_runs()
    n = ts.reduce(op, out, stream=stream, sync=True)
    st = ts.stats(); ts.close()
    return out, n, st

a, na, sta = run("sum", False)
b, nb, stb = run("sum", True)
print("kernels", sta["kernel"], stb["kernel"], "runs", na, nb)
bad = (a.value[:na].view(torch.int64) != b.value[:nb].view(torch.int64)).nonzero().flatten()
print("mismatching runs:", bad.numel())
cro = a.chrom_run_off.cpu().numpy()
for i in bad[:12].tolist():
    ch = int(np.searchsorted(cro, i, side="right") - 1)
    p = int(a.start[i].item())
    truth = 0.0
    for t in range(N):
        lo, hi = int(seg_off[ch * N + t]), int(seg_off[ch * N + t + 1])
        s = start[lo:hi]
        j = int(torch.searchsorted(s, torch.tensor([p], device=dev, dtype=torch.int32), right=True).item()) - 1
        if j >= 0 and int(finish[lo + j].item()) > p:
            truth += float(value[lo + j].item())
    print("run", i, "chrom", ch, "start", p, "finish", int(a.finish[i].item()), "delta", float(a.value[i]), "general", float(b.value[i]),
          "truth", truth, "window(4096)", (p - 1) // 4096, "pos_in_window", (p - 1) % 4096)
if bad.numel():
    d = bad[1:] - bad[:-1]
    print("index gaps between mismatches (first 20):", d[:20].tolist())

# dump the neighbourhood of the first mismatch for a CPU reproduction
if bad.numel():
    i = int(bad[0].item())
    ch = int(np.searchsorted(cro, i, side="right") - 1)
    p = int(a.start[i].item())
    lo_p, hi_p = p - 12000, p + 12000
    S, F, V, off = [], [], [], [0]
    for t in range(N):
        lo, hi = int(seg_off[ch * N + t]), int(seg_off[ch * N + t + 1])
        s, f, v = start[lo:hi], finish[lo:hi], value[lo:hi]
        m = (f > lo_p) & (s < hi_p)
        S.append(s[m].cpu().numpy()); F.append(f[m].cpu().numpy()); V.append(v[m].cpu().numpy())
        off.append(off[-1] + int(m.sum().item()))
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/mismatch_slice.npz", seg_off=np.array(off, np.int64), start=np.concatenate(S),
                        finish=np.concatenate(F), value=np.concatenate(V), p=p, chrom=ch,
                        first_start=int(min(int(start[int(seg_off[ch * N + t])].item()) for t in range(N))),
                        gen=b.value[i - 40:i + 200].cpu().numpy(), dlt=a.value[i - 40:i + 200].cpu().numpy(),
                        st=a.start[i - 40:i + 200].cpu().numpy())
    print("slice saved", off[-1], "intervals")
