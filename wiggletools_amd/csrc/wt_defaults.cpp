// wt_defaults.cpp -- default_value a reducer iterator advertises to its parent
// (what the reference computes once in each reducer's constructor).  Host only.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/wiggletools_amd.h"

namespace {
inline bool any_nan(const double *d, int n) {
    for (int i = 0; i < n; i++)
        if (std::isnan(d[i])) return true;
    return false;
}
inline double plain_sum(const double *d, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += d[i];
    return s;
}
inline double sq_error(const double *d, int n, double mean) {
    double e = 0;
    for (int i = 0; i < n; i++) e += (d[i] - mean) * (d[i] - mean);
    return e;
}
}  // namespace

extern "C" double wtamd_reducer_default(int op, int n, const double *d) {
    if (n <= 0 || !d) return NAN;
    const bool nan = any_nan(d, n);
    switch (op) {
    case WTAMD_OP_SUM:           // reducers.c:294-307
        return nan ? NAN : plain_sum(d, n);
    case WTAMD_OP_PRODUCT: {     // reducers.c:348-361
        if (nan) return NAN;
        double p = 1;
        for (int i = 0; i < n; i++) p *= d[i];
        return p;
    }
    case WTAMD_OP_MEAN: {        // reducers.c:404-422, stored through `float`
        if (nan) return NAN;
        const float f = (float) (plain_sum(d, n) / n);
        return f;
    }
    case WTAMD_OP_VAR: {         // reducers.c:481-505
        if (nan) return NAN;
        const double mean = plain_sum(d, n) / n;
        return sq_error(d, n, mean) / n;
    }
    case WTAMD_OP_STDDEV: {      // reducers.c:565-590
        if (nan) return NAN;
        const double mean = plain_sum(d, n) / n;
        return std::sqrt(sq_error(d, n, mean) / n);
    }
    case WTAMD_OP_ENTROPY: {     // reducers.c:640-663: `count / multi->count` is an int division
        if (nan) return NAN;
        int count = 0;
        for (int i = 0; i < n; i++) count += (d[i] != 0);
        const double p = count / n;
        return p ? -p * std::log(p) - (1 - p) * std::log(1 - p) : 0.0;
    }
    case WTAMD_OP_CV: {          // reducers.c:727-751, stored through `float`
        if (nan) return NAN;
        const double mean = plain_sum(d, n) / n;
        double e = 0;
        for (int i = 0; i < n; i++) e += (mean - d[i]) * (mean - d[i]);
        const float f = (float) (std::sqrt(e / n) / mean);
        return f;
    }
    case WTAMD_OP_MIN:           // reducers.c:237-253 (first element NaN is returned as is)
    case WTAMD_OP_MAX: {         // reducers.c:170-186
        if (std::isnan(d[0])) return d[0];
        if (nan) return NAN;
        return op == WTAMD_OP_MAX ? *std::max_element(d, d + n) : *std::min_element(d, d + n);
    }
    case WTAMD_OP_MEDIAN: {      // reducers.c:815-834, stored through `float`
        if (nan) return NAN;
        std::vector<double> tmp(d, d + n);
        std::sort(tmp.begin(), tmp.end());
        const float f = (float) tmp[n / 2];
        return f;
    }
    default:                     // ttest / MWU: NAN (setComparisons.c:130, 389)
        return NAN;
    }
}

// default_value of the operator iterator the reference builds around a track (host, like the constructors
// it restates; several store through `float`)
extern "C" double wtamd_map_default(int map_op, double param, double d) {
    const bool nan = d != d;
    switch (map_op) {
    case WTAMD_MAP_SCALE: { float f = nan ? NAN : d * param; return f; }                  // unaryOps.c:675-680
    case WTAMD_MAP_OFFSET: { float f = nan ? NAN : d + param; return f; }                 // :738-743
    case WTAMD_MAP_LN: return (!nan && d > 0) ? log(d) / 1.0 : NAN;                       // :792-796
    case WTAMD_MAP_LOG: return (!nan && d > 0) ? log(d) / log(param) : NAN;               // :807-811
    case WTAMD_MAP_EXP: { float f = nan ? NAN : exp(d * 1.0); return f; }                 // :860-865
    case WTAMD_MAP_EXPB: { float f = nan ? NAN : exp(d * log(param)); return f; }         // :847-852
    case WTAMD_MAP_POW: return (!nan && (d > 0 || param > 0)) ? pow(d, param) : NAN;      // :895-899
    case WTAMD_MAP_ABS: return nan ? NAN : fabs(d);
    case WTAMD_MAP_GT: case WTAMD_MAP_GTE: case WTAMD_MAP_LT: case WTAMD_MAP_LTE: return 0;   // :419
    default: return d;
    }
}

extern "C" {

// a := a (+) b on the host, b following a in genome order -- the same pairwise step the kernels use
void wtamd_pearson_merge(double *a, const double *b) {
    if (b[0] == 0) return;
    if (a[0] == 0) { memcpy(a, b, sizeof(double) * 6); return; }
    const double n = a[0] + b[0];
    const double dx = b[1] / b[0] - a[1] / a[0], dy = b[2] / b[0] - a[2] / a[0];
    const double w = a[0] * b[0] / n;
    a[3] += b[3] + dx * dx * w;
    a[4] += b[4] + dx * dy * w;
    a[5] += b[5] + dy * dy * w;
    a[0] = n; a[1] += b[1]; a[2] += b[2];
}

// T_XY / sqrt(T_XX T_YY), NaN when a track is constant (statistics.c:421-423: T_XX * T_YY == 0).  The reference's
// sequential update leaves EXACTLY 0 for a constant track; the same update applied slice by slice and merged leaves
// rounding noise of the order 1e-16 * n * mean^2 instead (the step subtracts two products of size mean^2), and
// noise / noise would be returned as a correlation: T below 1e-14 * n * mean^2 (a scatter under 1e-7 of the mean,
// the resolution of the float32 inputs) counts as constant.
double wtamd_pearson_finish(const double *m) {
    double txx = m[3], tyy = m[5];
    if (m[0] > 0) {
        const double mx = m[1] / m[0], my = m[2] / m[0];
        if (txx <= m[0] * mx * mx * 1e-14) txx = 0;
        if (tyy <= m[0] * my * my * 1e-14) tyy = 0;
    }
    const double den = txx * tyy;
    return den ? m[4] / sqrt(den) : __builtin_nan("");
}

}  // extern "C"
