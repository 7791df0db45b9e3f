// wt_mapop.h -- one `map`-able operator applied to one value (reference src/unaryOps.c, lines cited per
// case).  Shared by the device kernels (wt_map.hip) and by the host-side per-interval protocol of
// wtamd_MapIterator (wt_iter_abi.cpp: what a foreign consumer popping the iterator sees).
#pragma once
#include <cmath>

#include "../../include/wiggletools_amd.h"

#ifdef __HIPCC__
#define WM_FN __host__ __device__ inline
#else
#define WM_FN inline
#endif

// lg: log(param) for WTAMD_MAP_LOG / WTAMD_MAP_EXPB, 1.0 otherwise.  keep = false: the run is dropped.
WM_FN double wm_apply(int op, double param, double lg, double v, bool &keep) {
    keep = true;
    switch (op) {
    case WTAMD_MAP_SCALE: return (v != v) ? v : param * v;                       // unaryOps.c:650-664
    case WTAMD_MAP_OFFSET: return param + v;                                      // :722-734
    case WTAMD_MAP_LN:
    case WTAMD_MAP_LOG:                                                           // :760-779
        if (v <= 0) keep = false;
        return (v != v || v < 0) ? __builtin_nan("") : log(v) / lg;
    case WTAMD_MAP_EXP:
    case WTAMD_MAP_EXPB: return exp(v * lg);                                      // :823-835
    case WTAMD_MAP_POW: return ((param < 0 && v <= 0) || v != v) ? __builtin_nan("") : pow(v, param);   // :873-889
    case WTAMD_MAP_ABS: return (v != v) ? v : fabs(v);                            // :934-949
    case WTAMD_MAP_GT: keep = !(v <= param || v != v); return 1.0;                // :386-419, value stays 1
    case WTAMD_MAP_GTE: keep = !(v < param || v != v); return 1.0;
    case WTAMD_MAP_LT: keep = !(-1 * v <= -param || v != v); return 1.0;          // commandParser.c:185-189
    case WTAMD_MAP_LTE: keep = !(-1 * v < -param || v != v); return 1.0;
    default: return v;
    }
}
