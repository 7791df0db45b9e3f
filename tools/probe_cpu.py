"""How many host cores does this box really give us?  (cpu_count vs affinity vs cgroup quota vs a scaling probe)"""
import multiprocessing as mp
import os
import time


def burn(_):
    t0 = time.time()
    x = 0
    for i in range(6_000_000):
        x += i * i
    return time.time() - t0


if __name__ == "__main__":
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(p, open(p).read().strip())
        except Exception as e:
            print(p, "n/a")
    try:
        print(open("/proc/meminfo").read().split("\n")[0])
    except Exception:
        pass
    for w in (1, 4, 16, 64, 128, 256):
        with mp.get_context("spawn").Pool(w) as pool:
            pool.map(burn, range(w))        # warm
            t0 = time.time()
            r = pool.map(burn, range(w))
            wall = time.time() - t0
        print("workers %3d: wall %.2f s, mean task %.2f s, throughput %.1f tasks/s" % (w, wall, sum(r) / len(r), w / wall))
