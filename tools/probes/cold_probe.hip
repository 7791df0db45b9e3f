// What does the COLD run of a file-byte pipe pay for its memory, and what can be taken off the critical path?
// (round 5: the cold files -> result run spends 1.0-1.6 s obtaining 4 GB of page-locked staging and 13 GB of device memory)
//   1. hipHostMalloc vs (mmap + MADV_HUGEPAGE + pre-fault on T threads + hipHostRegister), per GiB
//   2. hipMalloc by size; hipMallocAsync from a pool
//   3. do allocations on a helper thread delay kernel launches / small copies of the main thread?
//   4. H2D bandwidth out of registered memory vs hipHostMalloc memory
// Build: hipcc --offload-arch=gfx950 -O2 -pthread tools/probes/cold_probe.hip -o tools/probes/cold_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void tiny(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

static void prefault(char *p, size_t n, int T) {
    std::vector<std::thread> th;
    const size_t per = (n / T + 4095) & ~(size_t) 4095;
    for (int t = 0; t < T; t++)
        th.emplace_back([=] {
            const size_t a = per * t, b = a + per < n ? a + per : n;
            for (size_t q = a; q < b; q += 4096) p[q] = 0;
        });
    for (auto &t : th) t.join();
}

int main() {
    hipSetDevice(0);
    int *d_ctr; hipMalloc(&d_ctr, 4); hipMemset(d_ctr, 0, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d_ctr); hipStreamSynchronize(s);
    const size_t G = (size_t) 1 << 30;

    // 1. page-locked memory
    for (int rep = 0; rep < 2; rep++) {
        void *h = nullptr;
        double t0 = now(); hipError_t e = hipHostMalloc(&h, G, hipHostMallocDefault); double t1 = now();
        hipHostFree(h); double t2 = now();
        printf("hipHostMalloc 1 GiB: %.1f ms (free %.1f ms) rc %d\n", t1 - t0, t2 - t1, (int) e);
    }
    for (int huge = 0; huge < 2; huge++)
        for (int T : {1, 4, 16}) {
            double t0 = now();
            char *p = (char *) mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, G, MADV_HUGEPAGE);
            double t1 = now();
            prefault(p, G, T);
            double t2 = now();
            hipError_t e = hipHostRegister(p, G, hipHostRegisterDefault);
            double t3 = now();
            void *dv = nullptr; hipMalloc(&dv, G);
            double t4 = now();
            hipMemcpyAsync(dv, p, G, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
            double t5 = now();
            hipFree(dv);
            hipHostUnregister(p); munmap(p, G);
            printf("mmap%s + prefault x%-2d %.1f ms + hipHostRegister %.1f ms = %.1f ms (rc %d); H2D of it %.1f GB/s\n", huge ? " (THP)" : "      ", T, t2 - t1, t3 - t2, t3 - t0,
                   (int) e, G / ((t5 - t4) * 1e-3) / 1e9);
        }
    {
        void *h = nullptr; hipHostMalloc(&h, G, hipHostMallocDefault);
        void *dv = nullptr; hipMalloc(&dv, G);
        double t4 = now(); hipMemcpyAsync(dv, h, G, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t5 = now();
        printf("H2D out of hipHostMalloc memory: %.1f GB/s\n", G / ((t5 - t4) * 1e-3) / 1e9);
        hipFree(dv); hipHostFree(h);
    }
    // registering in pieces (a ring of chunks): 64 x 16 MiB
    {
        char *p = (char *) mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, G, MADV_HUGEPAGE);
        prefault(p, G, 16);
        double t0 = now();
        for (size_t q = 0; q < G; q += (size_t) 16 << 20) hipHostRegister(p + q, (size_t) 16 << 20, hipHostRegisterDefault);
        double t1 = now();
        printf("hipHostRegister of 64 x 16 MiB pre-faulted: %.1f ms\n", t1 - t0);
        for (size_t q = 0; q < G; q += (size_t) 16 << 20) hipHostUnregister(p + q);
        munmap(p, G);
    }

    // 2. device memory
    for (size_t n : {G / 4, G, 4 * G}) {
        for (int rep = 0; rep < 2; rep++) {
            void *q = nullptr;
            double t0 = now(); hipError_t e = hipMalloc(&q, n); double t1 = now();
            hipFree(q); double t2 = now();
            printf("hipMalloc %5zu MiB: %.1f ms (free %.1f ms) rc %d\n", n >> 20, t1 - t0, t2 - t1, (int) e);
        }
    }
    {
        std::vector<void *> v;
        double t0 = now();
        for (int k = 0; k < 13; k++) { void *q = nullptr; hipMalloc(&q, G); v.push_back(q); }
        double t1 = now();
        printf("13 x hipMalloc 1 GiB back to back: %.1f ms\n", t1 - t0);
        for (void *q : v) hipFree(q);
    }
    {
        void *q = nullptr;
        double t0 = now(); hipError_t e = hipMallocAsync(&q, G, s); hipStreamSynchronize(s); double t1 = now();
        printf("hipMallocAsync 1 GiB: %.1f ms rc %d\n", t1 - t0, (int) e);
        if (e == hipSuccess) { hipFreeAsync(q, s); hipStreamSynchronize(s); }
        t0 = now(); e = hipMallocAsync(&q, G, s); hipStreamSynchronize(s); t1 = now();
        printf("hipMallocAsync 1 GiB again: %.1f ms rc %d\n", t1 - t0, (int) e);
        if (e == hipSuccess) { hipFreeAsync(q, s); hipStreamSynchronize(s); }
    }

    // 3. does a helper thread's allocating delay the main thread's launches?
    for (int mode = 0; mode < 4; mode++) {
        std::atomic<bool> go{false}, done{false};
        std::vector<void *> got;
        void *hgot = nullptr;
        double helper_ms = 0;
        std::thread helper([&] {
            hipSetDevice(0);
            while (!go.load()) { }
            double t0 = now();
            if (mode == 1) for (int k = 0; k < 8; k++) { void *q = nullptr; hipMalloc(&q, G); got.push_back(q); }
            if (mode == 2) hipHostMalloc(&hgot, 2 * G, hipHostMallocDefault);
            if (mode == 3) {
                char *p = (char *) mmap(nullptr, 2 * G, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                madvise(p, 2 * G, MADV_HUGEPAGE);
                prefault(p, 2 * G, 8);
                hipHostRegister(p, 2 * G, hipHostRegisterDefault);
                hgot = p;
            }
            helper_ms = now() - t0;
            done.store(true);
        });
        go.store(true);
        double worst = 0, sum = 0; int n = 0;
        double t_start = now();
        while (!done.load() || n < 200) {
            double t0 = now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d_ctr);
            hipStreamSynchronize(s);
            double dt = now() - t0;
            if (dt > worst) worst = dt;
            sum += dt; n++;
            if (now() - t_start > 20000) break;
        }
        helper.join();
        printf("main thread launch+sync while helper %s: mean %.3f ms, worst %.2f ms over %d launches; helper took %.1f ms\n",
               mode == 0 ? "idles" : mode == 1 ? "hipMallocs 8 x 1 GiB" : mode == 2 ? "hipHostMallocs 2 GiB" : "mmap+prefault x8+registers 2 GiB", sum / n, worst, n, helper_ms);
        for (void *q : got) hipFree(q);
        if (mode == 2 && hgot) hipHostFree(hgot);
        if (mode == 3 && hgot) { hipHostUnregister(hgot); munmap(hgot, 2 * G); }
    }
    return 0;
}
