/*
 * wt_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's Multiplexer -> reducer path
 * (WiggleTools v1.2.11).  It is the checker the HIP kernels are compared
 * against; nothing in the product (wiggletools_amd/) links or calls it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it.
 *
 * What it follows (reference file:line under /root/reference/src):
 *   alignment ......... multiplexer.c:37-128  (popCoreMultiplexer2 and helpers)
 *   two-set alignment . multiSet.c:21-101, setComparisons.c:48-54,282-288
 *   sum/product/mean .. reducers.c:259-292, 313-346, 367-402
 *   var/stddev/CV ..... reducers.c:428-479, 511-563, 672-725
 *   entropy ........... reducers.c:665 (installs StdDevReductionPop)
 *   min/max ........... reducers.c:192-235, 125-168
 *   median ............ reducers.c:780-813
 *   t-test ............ setComparisons.c:35-121
 *   Mann-Whitney U .... setComparisons.c:269-370, ctor :372-390
 *   reducer defaults .. the ctor of each op in reducers.c
 *   AUC ............... statistics.c:103-120
 *   compression ....... unaryOps.c:235-253
 *
 * Pinning: every reducers.c op and the alignment are checked against the
 * compiled reference itself (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle_vs_ref.py) and against the reference fixtures' golden
 * vectors (tests/golden/).  setComparisons.c cannot be compiled in this image
 * (it includes <gsl/gsl_cdf.h>; GSL is absent and un-vendored), therefore:
 *   - MWU is pinned only by the golden vector the survey captured from the
 *     reference binary -> "parity partially pinned";
 *   - the t-test statistic (t, nu) follows the source; the final Student-t
 *     tail (gsl_cdf_tdist_Q, GSL version unpinned, Ubuntu 20.04 => 2.5) is
 *     restated from the published definition Q(t;nu) = I_x(nu/2,1/2)/2,
 *     x = nu/(nu+t^2), and cross-checked against scipy.stats.t.sf in this
 *     container -> "parity unpinned" at that boundary.
 *
 * The alignment here deliberately uses linear scans over the N tracks instead
 * of the reference's Fibonacci heaps: same run sequence (verified against
 * oracle/_ref), much simpler to audit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum {
    OP_SUM = 0, OP_PRODUCT, OP_MEAN, OP_VAR, OP_STDDEV, OP_ENTROPY, OP_CV,
    OP_MIN, OP_MAX, OP_MEDIAN, OP_TTEST, OP_MWU, OP_COUNT_
};
#define STRICT_SET0 1u
#define STRICT_SET1 2u

typedef struct {
    int32_t n_chrom, n_tracks;
    const int64_t *seg_off;      /* n_chrom*n_tracks+1 */
    const int32_t *start, *finish;
    const double *value;
    const double *defaults;      /* n_tracks */
} wto_tracks;

/* ------------------------------------------------------------------ */
/* Per-run reducers. v[i] = in-play value, inplay[i] flag, dflt[i].    */
/* ------------------------------------------------------------------ */

static double pick(const double *v, const char *inplay, const double *dflt, int i) {
    return inplay[i] ? v[i] : dflt[i];
}

/* reducers.c:259-292 */
static double red_sum(int n, const double *v, const char *ip, const double *d) {
    double acc = 0;
    for (int i = 0; i < n; i++) {
        double x = pick(v, ip, d, i);
        if (isnan(x)) return NAN;
        acc += x;
    }
    return acc;
}

/* reducers.c:313-346 */
static double red_product(int n, const double *v, const char *ip, const double *d) {
    double acc = 1;
    for (int i = 0; i < n; i++) {
        double x = pick(v, ip, d, i);
        if (isnan(x)) return NAN;
        acc *= x;
    }
    return acc;
}

/* reducers.c:367-402 : divides by the track count, not the in-play count */
static double red_mean(int n, const double *v, const char *ip, const double *d) {
    double s = red_sum(n, v, ip, d);
    if (!isnan(s)) s /= n;
    return s;
}

/* reducers.c:125-168 / 192-235 : seed is 0 (not the default) if track 0 is absent */
static double red_minmax(int n, const double *v, const char *ip, const double *d, int want_max) {
    double best = ip[0] ? v[0] : 0;
    if (isnan(best)) return best;
    for (int i = 1; i < n; i++) {
        double x = pick(v, ip, d, i);
        if (isnan(x)) return NAN;
        if (want_max ? (best < x) : (best > x)) best = x;
    }
    return best;
}

/* reducers.c:428-479 : float-rounded pass 1, pass 2 over in-play tracks only */
static double red_var(int n, const double *v, const char *ip, const double *d) {
    double mean = 0, count = 0;
    for (int i = 0; i < n; i++) {
        float x = (float) pick(v, ip, d, i);
        if (isnan(x)) { mean = NAN; break; }
        mean += x;
        count++;
    }
    if (count < 2 || isnan(mean)) return NAN;
    mean /= count;
    double acc = 0;
    for (int i = 0; i < n; i++) {
        if (ip[i]) {
            double diff = mean - v[i];
            acc += diff * diff;
        }
    }
    return acc / count;
}

/* reducers.c:511-563 (also the pop the `entropy` ctor installs, :665) */
static double red_stddev(int n, const double *v, const char *ip, const double *d) {
    double mean = 0;
    for (int i = 0; i < n; i++) {
        float x = (float) pick(v, ip, d, i);
        if (isnan(x)) return NAN;
        mean += x;
    }
    if (isnan(mean)) return NAN;
    mean /= n;
    double acc = 0;
    for (int i = 0; i < n; i++) {
        double diff = mean - pick(v, ip, d, i);
        acc += diff * diff;
    }
    acc /= n;
    return sqrt(acc);
}

/* reducers.c:672-725 */
static double red_cv(int n, const double *v, const char *ip, const double *d) {
    double mean = 0;
    for (int i = 0; i < n; i++) {
        float x = (float) pick(v, ip, d, i);
        mean += x;
        if (isnan(x)) return NAN;
    }
    mean /= n;
    if (mean == 0) return NAN;
    double acc = 0;
    for (int i = 0; i < n; i++) {
        double diff = mean - pick(v, ip, d, i);
        acc += diff * diff;
    }
    acc /= n;
    return sqrt(acc) / mean;
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *) a, y = *(const double *) b;
    return (x < y) ? -1 : (x > y) ? 1 : 0;
}

/* reducers.c:780-813 : upper median vals[n/2], any NaN -> NaN */
static double red_median(int n, const double *v, const char *ip, const double *d, double *scratch) {
    for (int i = 0; i < n; i++) {
        scratch[i] = pick(v, ip, d, i);
        if (isnan(scratch[i])) return NAN;
    }
    qsort(scratch, n, sizeof(double), cmp_double);
    return scratch[n / 2];
}

/* ------------------------------------------------------------------ */
/* Student-t upper tail (stands in for gsl_cdf_tdist_Q, see header).   */
/* ------------------------------------------------------------------ */

static double betacf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    double qab = a + b, qap = a + 1, qam = a - 1;
    double c = 1, dd = 1 - qab * x / qap;
    if (fabs(dd) < tiny) dd = tiny;
    dd = 1 / dd;
    double h = dd;
    for (int m = 1; m <= 10000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        dd = 1 + aa * dd; if (fabs(dd) < tiny) dd = tiny;
        c = 1 + aa / c;   if (fabs(c) < tiny) c = tiny;
        dd = 1 / dd; h *= dd * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        dd = 1 + aa * dd; if (fabs(dd) < tiny) dd = tiny;
        c = 1 + aa / c;   if (fabs(c) < tiny) c = tiny;
        dd = 1 / dd;
        double del = dd * c;
        h *= del;
        if (fabs(del - 1) < eps) break;
    }
    return h;
}

/* y = 1 - x, handed over by the caller who can form it WITHOUT the cancellation (round 6: with 1 - x of the rounded
 * x = nu / (nu + t^2) a small t -- p close to 1 -- lost all but a few digits of 1 - p: t = 2e-5, nu = 898 came out 1e-9
 * off; gsl_cdf_tdist_Q, which this stands in for, has no such loss) */
static double inc_beta(double a, double b, double x, double y) {
    if (x <= 0) return 0;
    if (y <= 0) return 1;
    double lnfront = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(y);
    if (x < (a + 1) / (a + b + 2))
        return exp(lnfront) * betacf(a, b, x) / a;
    return 1 - exp(lnfront) * betacf(b, a, y) / b;
}

double wto_tdist_Q(double t, double nu) {
    if (isnan(t) || isnan(nu) || nu <= 0) return NAN;
    if (isinf(t)) return t > 0 ? 0 : 1;
    double x = nu / (nu + t * t), y = t * t / (nu + t * t);
    double tail = 0.5 * inc_beta(nu / 2, 0.5, x, y);
    return t >= 0 ? tail : 1 - tail;
}

/* setComparisons.c:60-117. Sums over in-play tracks, counts over all tracks. */
static double red_ttest(int n1, int n2, const double *v, const char *ip) {
    double s1 = 0, s2 = 0, q1 = 0, q2 = 0;
    for (int i = 0; i < n1; i++)
        if (ip[i]) { s1 += v[i]; q1 += v[i] * v[i]; }
    for (int i = n1; i < n1 + n2; i++)
        if (ip[i]) { s2 += v[i]; q2 += v[i] * v[i]; }
    if (n1 == 0 || n2 == 0) return NAN;
    double m1 = s1 / n1, m2 = s2 / n2;
    double msq1 = q1 / n1, msq2 = q2 / n2;
    double var1 = msq1 - m1 * m1, var2 = msq2 - m2 * m2;
    if (var1 + var2 == 0) return NAN;
    double t = (m1 - m2) / sqrt(var1 / n1 + var2 / n2);
    if (t < 0) t = -t;
    double den = (var1 / n1 + var2 / n2);
    double c1 = (double) ((int64_t) n1 * n1 * (n1 - 1));
    double c2 = (double) ((int64_t) n2 * n2 * (n2 - 1));
    double nu = den * den / ((var1 * var1) / c1 + (var2 * var2) / c2);
    return 2 * wto_tdist_Q(t, nu);
}

/* Exposes (t, nu) for tests that pin the statistic separately from the tail. */
void wto_ttest_stat(int n1, int n2, const double *v, const char *ip, double *t_out, double *nu_out) {
    double s1 = 0, s2 = 0, q1 = 0, q2 = 0;
    for (int i = 0; i < n1; i++)
        if (ip[i]) { s1 += v[i]; q1 += v[i] * v[i]; }
    for (int i = n1; i < n1 + n2; i++)
        if (ip[i]) { s2 += v[i]; q2 += v[i] * v[i]; }
    double m1 = s1 / n1, m2 = s2 / n2;
    double var1 = q1 / n1 - m1 * m1, var2 = q2 / n2 - m2 * m2;
    double den = (var1 / n1 + var2 / n2);
    double t = fabs((m1 - m2) / sqrt(den));
    double c1 = (double) ((int64_t) n1 * n1 * (n1 - 1));
    double c2 = (double) ((int64_t) n2 * n2 * (n2 - 1));
    *t_out = t;
    *nu_out = den * den / ((var1 * var1) / c1 + (var2 * var2) / c2);
}

typedef struct { double value; char set; } vsp;

/* Stable merge sort by value: glibc 2.35 qsort (merge sort when the scratch
 * fits) keeps insertion order inside a tie group, i.e. set-0 before set-1
 * (setComparisons.c:297-325). */
static void vsp_sort(vsp *a, vsp *tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    vsp_sort(a, tmp, h);
    vsp_sort(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].value < a[i].value) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t) n * sizeof(vsp));
}

/* setComparisons.c:293-366 with mu/sigma from the ctor :386-387 (C int division). */
static double red_mwu(int n1, int n2, const double *v, const char *ip, const double *d, vsp *tab, vsp *tmp) {
    int N = n1 + n2;
    for (int i = 0; i < N; i++) {
        tab[i].value = pick(v, ip, d, i);
        if (isnan(tab[i].value)) return NAN;
        tab[i].set = (i >= n1);
    }
    vsp_sort(tab, tmp, N);
    /* setComparisons.c:386-387: C int products (undefined once they overflow, ~1300 tracks per set; here they wrap as the
     * engine's do, csrc/wt_plan.h wt_mwu_make_table) and int divisions */
    const int n12 = (int) ((unsigned) n1 * (unsigned) n2);
    double mu = (double) (n12 / 2);
    double sigma = sqrt((double) ((int) ((unsigned) n12 * ((unsigned) n1 + (unsigned) n2 + 1u)) / 12));
    double U1 = 0;
    int prev = 0, ties = 0, prevTies = 0;
    for (int idx = 0; idx < N && prev < n1; idx++) {
        if (!tab[idx].set) {
            U1 += idx - prev;
            if (ties) {
                int j;
                for (j = idx + 1; j < N && tab[j].value == tab[idx].value && tab[j].set; j++)
                    prevTies++;
                U1 -= prevTies / 2.0;
                U1 += (ties - prevTies) / 2.0;
                if (prevTies == ties) prevTies = ties = 0;
            } else {
                int j;
                for (j = idx + 1; j < N && tab[j].value == tab[idx].value; j++)
                    if (tab[j].set) ties++;
                if (ties) U1 += ties / 2.0;
            }
            prev++;
        }
    }
    if (U1 > mu) return 2 * erf((mu - U1) / sigma);
    return 2 * erf((U1 - mu) / sigma);
}

/* ------------------------------------------------------------------ */
/* Reducer ctor defaults (what the reducer advertises to its parent).  */
/* ------------------------------------------------------------------ */
double wto_reducer_default(int op, int n, const double *d) {
    int i;
    switch (op) {
    case OP_SUM: {          /* reducers.c:294-307 */
        double s = 0;
        for (i = 0; i < n; i++) { if (isnan(d[i])) return NAN; s += d[i]; }
        return s;
    }
    case OP_PRODUCT: {      /* reducers.c:348-361 */
        double p = 1;
        for (i = 0; i < n; i++) { if (isnan(d[i])) return NAN; p *= d[i]; }
        return p;
    }
    case OP_MEAN: {         /* reducers.c:404-422 : through `float` */
        double s = 0;
        for (i = 0; i < n; i++) { if (isnan(d[i])) { s = NAN; break; } s += d[i]; }
        float f = isnan(s) ? NAN : (float) (s / n);
        return f;
    }
    case OP_VAR: case OP_STDDEV: {   /* reducers.c:481-505, 565-590 */
        double s = 0;
        for (i = 0; i < n; i++) { if (isnan(d[i])) return NAN; s += d[i]; }
        double mean = s / n, err = 0;
        for (i = 0; i < n; i++) err += (d[i] - mean) * (d[i] - mean);
        return op == OP_VAR ? err / n : sqrt(err / n);
    }
    case OP_ENTROPY: {      /* reducers.c:640-663 : integer division count/n */
        int count = 0;
        for (i = 0; i < n; i++) { if (isnan(d[i])) return NAN; if (d[i] != 0) count++; }
        double p = count / n;
        return p ? -p * log(p) - (1 - p) * log(1 - p) : 0;
    }
    case OP_CV: {           /* reducers.c:727-751 : through `float` */
        double mean = 0;
        for (i = 0; i < n; i++) { if (isnan(d[i])) { mean = NAN; break; } mean += d[i]; }
        if (isnan(mean)) return NAN;
        mean /= n;
        double err = 0;
        for (i = 0; i < n; i++) err += (mean - d[i]) * (mean - d[i]);
        float f = (float) (sqrt(err / n) / mean);
        return f;
    }
    case OP_MIN: case OP_MAX: {   /* reducers.c:170-186, 237-253 */
        double b = d[0];
        if (isnan(b)) return b;
        for (i = 1; i < n; i++) {
            if (isnan(d[i])) return NAN;
            if (op == OP_MAX ? d[i] > b : d[i] < b) b = d[i];
        }
        return b;
    }
    case OP_MEDIAN: {       /* reducers.c:815-834 : through `float` */
        double *tmp = (double *) malloc(sizeof(double) * (size_t) n);
        float f = 0;
        for (i = 0; i < n; i++) { tmp[i] = d[i]; if (isnan(d[i])) { f = NAN; break; } }
        if (!isnan(f)) { qsort(tmp, n, sizeof(double), cmp_double); f = (float) tmp[n / 2]; }
        free(tmp);
        return f;
    }
    default:                /* ttest / MWU: NAN (setComparisons.c:130,389) */
        return NAN;
    }
}

/* ------------------------------------------------------------------ */
/* Alignment sweep (multiplexer.c:98-121 without heaps).               */
/* ------------------------------------------------------------------ */

typedef struct {
    int op; unsigned flags; int n_set0;
    int64_t cap, n;
    int32_t *o_chrom, *o_start, *o_finish;
    double *o_value;
    double *o_tile; uint8_t *o_inplay;   /* optional materialised tile */
} sink;

static int set_in_play(const char *ip, int lo, int hi, int strict) {
    int c = 0;
    for (int i = lo; i < hi; i++) c += ip[i] != 0;
    return strict ? (c == hi - lo) : (c > 0);
}

/* Returns runs emitted, or -1 if the capacity was exceeded. */
static int64_t sweep(const wto_tracks *t, sink *s) {
    const int N = t->n_tracks;
    int64_t *cur = (int64_t *) malloc(sizeof(int64_t) * (size_t) N);
    int64_t *end = (int64_t *) malloc(sizeof(int64_t) * (size_t) N);
    char *ip = (char *) calloc((size_t) N, 1);
    double *v = (double *) malloc(sizeof(double) * (size_t) N);
    double *scratch = (double *) malloc(sizeof(double) * (size_t) N);
    vsp *tab = (vsp *) malloc(sizeof(vsp) * (size_t) N), *tmp = (vsp *) malloc(sizeof(vsp) * (size_t) N);
    const int two = (s->op == OP_TTEST || s->op == OP_MWU);
    const int n1 = two ? s->n_set0 : N, n2 = N - n1;
    int64_t overflow = 0;

    for (int c = 0; c < t->n_chrom && !overflow; c++) {
        for (int i = 0; i < N; i++) {
            cur[i] = t->seg_off[(int64_t) c * N + i];
            end[i] = t->seg_off[(int64_t) c * N + i + 1];
            ip[i] = 0;
            v[i] = t->defaults[i];
        }
        int inplay_count = 0;
        int32_t start = 0, finish = 0;
        for (;;) {
            /* popClosingWiggleIterators, multiplexer.c:37-48 */
            for (int i = 0; i < N; i++)
                if (ip[i] && t->finish[cur[i]] == finish) {
                    cur[i]++; ip[i] = 0; inplay_count--; v[i] = t->defaults[i];
                }
            /* anything left on this chromosome? (queueUp..., :50-74) */
            int waiting = 0;
            int32_t min_start = INT32_MAX;
            for (int i = 0; i < N; i++)
                if (!ip[i] && cur[i] < end[i]) {
                    waiting = 1;
                    if (t->start[cur[i]] < min_start) min_start = t->start[cur[i]];
                }
            if (!inplay_count && !waiting) break;
            /* :112-115 */
            start = inplay_count ? finish : min_start;
            /* admitNewWiggleIteratorsIntoPlay, :76-85 */
            for (int i = 0; i < N; i++)
                if (!ip[i] && cur[i] < end[i] && t->start[cur[i]] == start) {
                    ip[i] = 1; inplay_count++; v[i] = t->value[cur[i]];
                }
            /* defineNewFinish, :87-96 */
            finish = INT32_MAX;
            for (int i = 0; i < N; i++) {
                if (ip[i]) { if (t->finish[cur[i]] < finish) finish = t->finish[cur[i]]; }
                else if (cur[i] < end[i] && t->start[cur[i]] < finish) finish = t->start[cur[i]];
            }
            /* emission predicate: strict (:120,125) / both sets (setComparisons.c:48-54) */
            int emit;
            if (two)
                emit = set_in_play(ip, 0, n1, s->flags & STRICT_SET0) &&
                       set_in_play(ip, n1, N, s->flags & STRICT_SET1);
            else
                emit = (s->flags & STRICT_SET0) ? (inplay_count == N) : 1;
            if (!emit) continue;
            if (s->n >= s->cap) { overflow = 1; break; }
            double r = NAN;
            switch (s->op) {
            case OP_SUM: r = red_sum(N, v, ip, t->defaults); break;
            case OP_PRODUCT: r = red_product(N, v, ip, t->defaults); break;
            case OP_MEAN: r = red_mean(N, v, ip, t->defaults); break;
            case OP_VAR: r = red_var(N, v, ip, t->defaults); break;
            case OP_STDDEV: case OP_ENTROPY: r = red_stddev(N, v, ip, t->defaults); break;
            case OP_CV: r = red_cv(N, v, ip, t->defaults); break;
            case OP_MIN: r = red_minmax(N, v, ip, t->defaults, 0); break;
            case OP_MAX: r = red_minmax(N, v, ip, t->defaults, 1); break;
            case OP_MEDIAN: r = red_median(N, v, ip, t->defaults, scratch); break;
            case OP_TTEST: r = red_ttest(n1, n2, v, ip); break;
            case OP_MWU: r = red_mwu(n1, n2, v, ip, t->defaults, tab, tmp); break;
            default: break;
            }
            s->o_chrom[s->n] = c; s->o_start[s->n] = start; s->o_finish[s->n] = finish;
            if (s->o_value) s->o_value[s->n] = r;
            if (s->o_tile)
                for (int i = 0; i < N; i++) {
                    s->o_tile[s->n * N + i] = v[i];
                    s->o_inplay[s->n * N + i] = (uint8_t) ip[i];
                }
            s->n++;
        }
    }
    free(cur); free(end); free(ip); free(v); free(scratch); free(tab); free(tmp);
    return overflow ? -1 : s->n;
}

int64_t wto_reduce(const wto_tracks *t, int op, unsigned flags, int n_set0, int64_t cap,
                   int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    sink s = { op, flags, n_set0, cap, 0, o_chrom, o_start, o_finish, o_value, NULL, NULL };
    return sweep(t, &s);
}

/* Run list + the values[]/inplay[] tile a Multiplexer exposes (multiplexer.h:21-36). */
int64_t wto_multiplex(const wto_tracks *t, unsigned flags, int64_t cap,
                      int32_t *o_chrom, int32_t *o_start, int32_t *o_finish,
                      double *o_tile, uint8_t *o_inplay) {
    sink s = { -1, flags, 0, cap, 0, o_chrom, o_start, o_finish, NULL, o_tile, o_inplay };
    return sweep(t, &s);
}

/* statistics.c:103-120 */
double wto_auc(int64_t n, const int32_t *start, const int32_t *finish, const double *value) {
    double res = 0;
    for (int64_t r = 0; r < n; r++)
        if (!isnan(value[r])) res += (finish[r] - start[r]) * value[r];
    return res;
}

/* statistics.c:414-458 PearsonPop over the Multiplexer tile of tracks {0,1}: x[r], y[r] are the
 * default-substituted values of run r.  The reference's online update, statement for statement;
 * `count` is a C int there (overflows past 2^31 covered bp -- not reproduced: int64 here).
 * Result NaN unless T_XX*T_YY != 0 (:421-423). */
double wto_pearson(int64_t n, const int32_t *start, const int32_t *finish, const double *x, const double *y) {
    int64_t count = 0;
    double sum_X = 0, sum_Y = 0, T_XX = 0, T_XY = 0, T_YY = 0;
    for (int64_t r = 0; r < n; r++) {
        const double X = x[r], Y = y[r];
        const int length = finish[r] - start[r];
        if (count) {
            double old_mean_X = sum_X / count;
            double new_mean_X = sum_X / (count + length);
            double old_mean_Y = sum_Y / count;
            double new_mean_Y = sum_Y / (count + length);
            double scaling_ratio = (double) count / (count + length);
            T_XY += (new_mean_X * old_mean_Y + scaling_ratio * X * Y - new_mean_X * Y - new_mean_Y * X) * length;
            T_XX += (new_mean_X * (old_mean_X - 2 * X) + scaling_ratio * X * X) * length;
            T_YY += (new_mean_Y * (old_mean_Y - 2 * Y) + scaling_ratio * Y * Y) * length;
        }
        count += length;
        sum_X += X * length;
        sum_Y += Y * length;
    }
    if (T_XX * T_YY) return T_XY / sqrt(T_XX * T_YY);
    return NAN;
}

/* `map`-able unary operators, value by value (unaryOps.c: scale :650-664, offset :722-734, ln / log
 * :760-779 with ctor :786-813, exp :823-835 with ctors :843-866, pow :873-889, abs :934-949).
 * out[i] = f(in[i]); keep[i] = 0 for intervals the operator skips (ln / log: value <= 0, :764-765).
 * gt / gte / lt / lte (commandParser.c:180-199 -> HighPassFilterWiggleIterator, unaryOps.c:386-419,
 * lt / lte through ScaleWiggleIterator(-1)): runs failing the comparison and NaN runs are skipped,
 * the kept ones carry the iterator's initial value 1 (wiggleIterator.c:26).
 * map_op: 0 scale, 1 offset, 2 ln, 3 log base param, 4 exp, 5 exp radix param, 6 pow, 7 abs,
 * 8 gt, 9 gte, 10 lt, 11 lte. */
void wto_map(int map_op, double param, int64_t n, const double *in, double *out, unsigned char *keep) {
    const double lg = (map_op == 3 || map_op == 5) ? log(param) : 1.0;     /* baseLog / radixLog */
    for (int64_t i = 0; i < n; i++) {
        const double v = in[i];
        double r = v;
        int k = 1;
        switch (map_op) {
        case 0: r = isnan(v) ? NAN : param * v; break;
        case 1: r = param + v; break;
        case 2: case 3:
            if (v <= 0) k = 0;                                  /* skipped (NaN <= 0 is false: kept) */
            r = (isnan(v) || v < 0) ? NAN : log(v) / lg;
            break;
        case 4: case 5: r = exp(v * lg); break;
        case 6: r = ((param < 0 && v <= 0) || isnan(v)) ? NAN : pow(v, param); break;
        case 7: r = isnan(v) ? NAN : fabs(v); break;
        case 8: k = !(v <= param || isnan(v)); r = 1; break;
        case 9: k = !(v < param || isnan(v)); r = 1; break;
        case 10: k = !(-1 * v <= -param || isnan(v)); r = 1; break;     /* scale(-1), then gt -param */
        case 11: k = !(-1 * v < -param || isnan(v)); r = 1; break;
        default: break;
        }
        out[i] = r;
        if (keep) keep[i] = (unsigned char) k;
    }
}

/* default_value of the operator iterator: several ctors store it through `float` */
double wto_map_default(int map_op, double param, double d) {
    switch (map_op) {
    case 0: { float f = isnan(d) ? NAN : d * param; return f; }             /* :675-680 float */
    case 1: { float f = isnan(d) ? NAN : d + param; return f; }             /* :738-743 float */
    case 2: return (!isnan(d) && d > 0) ? log(d) / 1.0 : NAN;                /* :792-796 double */
    case 3: return (!isnan(d) && d > 0) ? log(d) / log(param) : NAN;         /* :807-811 double */
    case 4: { float f = isnan(d) ? NAN : exp(d * 1.0); return f; }           /* :860-865 float */
    case 5: { float f = isnan(d) ? NAN : exp(d * log(param)); return f; }    /* :847-852 float */
    case 6: return (!isnan(d) && (d > 0 || param > 0)) ? pow(d, param) : NAN;/* :895-899 double */
    case 7: return isnan(d) ? NAN : fabs(d);
    case 8: case 9: case 10: case 11: return 0;                              /* unaryOps.c:419 */
    default: return d;
    }
}

/* unaryOps.c:235-253 : in-place merge of adjacent runs |dv| < 1e-6 (or both NaN). Returns new count. */
int64_t wto_compress(int64_t n, int32_t *chrom, int32_t *start, int32_t *finish, double *value) {
    int64_t w = 0;
    for (int64_t r = 0; r < n; r++) {
        if (w > 0 && chrom[r] == chrom[w - 1] && start[r] == finish[w - 1] &&
            ((isnan(value[r]) && isnan(value[w - 1])) || fabs(value[r] - value[w - 1]) < 0.000001)) {
            finish[w - 1] = finish[r];
        } else {
            chrom[w] = chrom[r]; start[w] = start[r]; finish[w] = finish[r]; value[w] = value[r];
            w++;
        }
    }
    return w;
}
