// wt_devscope.h -- temporary device buffers of one host entry point, released on every exit path
// (the WT_HIP-style early returns included).
#ifndef WT_DEVSCOPE_H_
#define WT_DEVSCOPE_H_
#include <hip/hip_runtime.h>

#include <vector>

struct WtDevScope {
    std::vector<void *> ptrs;
    WtDevScope() = default;
    WtDevScope(const WtDevScope &) = delete;
    WtDevScope &operator=(const WtDevScope &) = delete;
    template <class T>
    hipError_t alloc(T **p, size_t bytes) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
        if (e == hipSuccess) { ptrs.push_back(q); *p = (T *) q; }
        return e;
    }
    ~WtDevScope() {
        for (void *q : ptrs) (void) hipFree(q);
    }
};

#endif  // WT_DEVSCOPE_H_
