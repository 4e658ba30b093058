// Vector algebra of the batched scaled-conjugate-gradient driver: the passes over the
// parameter-sized arrays that one SCG iteration (auxiliary/optimize.py:132-238) makes
// between its two gradient evaluations, for P independent problems at once.
//
// A "parameter array" holds, for every problem, the n = nA + nb numbers of (A(t), b(t)).
// It is stored as two contiguous blocks, [P x nA | P x nb], so that the A block and the
// b block are exactly the (P,N,D,D) and (P,N,D) arrays the evaluation kernels take and
// no packing copy is ever made.  Every problem carries its own step sizes and its own
// accept / reject state: each pass takes a per-problem `mode` (0 = leave this problem
// alone, no memory traffic) and per-problem scalars.
//
// Reductions are two-stage and deterministic: every block reduces one chunk of one
// problem's row into work[((p*2 + z)*nch + chunk)*4 + k], a second kernel adds the
// chunks in index order.
#include "common.h"

// Products and sums round separately in this file (the element-wise results are then the
// same bits a two-operation formulation gives); the passes are bound by memory traffic.
#pragma clang fp contract(off)

namespace {

constexpr int NT = 256;   // threads per block
constexpr int U = 4;      // independent units in flight per thread

struct Geom {
    int P;
    long long nA, nb;
};

__device__ inline size_t seg_offset(const Geom& g, int p, int z) {
    return z ? (size_t)g.P * (size_t)g.nA + (size_t)p * (size_t)g.nb : (size_t)p * (size_t)g.nA;
}

template <int V> struct Vec;
template <> struct Vec<1> {
    double v[1];
    __device__ static Vec load(const double* p) { Vec r; r.v[0] = *p; return r; }
    __device__ void store(double* p) const { *p = v[0]; }
};
template <> struct Vec<2> {
    double v[2];
    __device__ static Vec load(const double* p) {
        const double2 t = *reinterpret_cast<const double2*>(p);
        Vec r; r.v[0] = t.x; r.v[1] = t.y; return r;
    }
    __device__ void store(double* p) const { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); }
};

// ---------------------------------------------------------------------------------
// The passes.  Each one: K results per problem (bit k of MAXMASK: result k is a
// maximum, otherwise a sum), `In` = what load() fetches for one unit of V consecutive
// numbers at offset `at`, and apply() = compute, accumulate and store for that unit.
// ---------------------------------------------------------------------------------

// {d.g, d.d, g.g}: slope and squared length of the direction (optimize.py:150-156)
struct Dots3 {
    static constexpr int K = 3;
    static constexpr unsigned MAXMASK = 0;
    const double *d, *g;
    template <int V> struct In { Vec<V> d, g; };
    template <int V> __device__ In<V> load(size_t at, int) const { return {Vec<V>::load(d + at), Vec<V>::load(g + at)}; }
    template <int V> __device__ void apply(const In<V>& in, size_t, int, int, double (&acc)[4]) const {
        for (int i = 0; i < V; ++i) {
            acc[0] += in.d.v[i] * in.g.v[i];
            acc[1] += in.d.v[i] * in.d.v[i];
            acc[2] += in.g.v[i] * in.g.v[i];
        }
    }
};

// y = x + coef*dir with dir = d (mode 1) or dir = -g, also stored as the new d (mode 2:
// the direction pointed uphill and is replaced by steepest descent, optimize.py:151-153).
// Used for the probe point x + sigma d (:158) and the trial point x + alpha d (:170).
struct Step {
    static constexpr int K = 0;
    static constexpr unsigned MAXMASK = 0;
    const double *x, *g, *coef;
    double *d, *y;
    template <int V> struct In { Vec<V> x, dir; };
    template <int V> __device__ In<V> load(size_t at, int md) const {
        return {Vec<V>::load(x + at), Vec<V>::load((md == 2 ? g : d) + at)};
    }
    template <int V> __device__ void apply(const In<V>& in, size_t at, int p, int md, double (&)[4]) const {
        const double c = coef[p];
        Vec<V> dir, out;
        for (int i = 0; i < V; ++i) {
            dir.v[i] = md == 2 ? -in.dir.v[i] : in.dir.v[i];
            out.v[i] = in.x.v[i] + c * dir.v[i];
        }
        out.store(y + at);
        if (md == 2) dir.store(d + at);
    }
};

// d.(g_probe - g): numerator of the curvature estimate theta (optimize.py:159)
struct Curvature {
    static constexpr int K = 1;
    static constexpr unsigned MAXMASK = 0;
    const double *d, *gp, *g;
    template <int V> struct In { Vec<V> d, gp, g; };
    template <int V> __device__ In<V> load(size_t at, int) const {
        return {Vec<V>::load(d + at), Vec<V>::load(gp + at), Vec<V>::load(g + at)};
    }
    template <int V> __device__ void apply(const In<V>& in, size_t, int, int, double (&acc)[4]) const {
        for (int i = 0; i < V; ++i) acc[0] += in.d.v[i] * (in.gp.v[i] - in.g.v[i]);
    }
};

// For an accepted trial point: {max|alpha d| (:186), g_try.g_try (:196),
// g_try.(g - g_try) = numerator of the Polak-Ribiere factor (:221), sum|g_try|}
struct TrialStats {
    static constexpr int K = 4;
    static constexpr unsigned MAXMASK = 1;
    const double *d, *g, *gt, *alpha;
    template <int V> struct In { Vec<V> d, g, gt; };
    template <int V> __device__ In<V> load(size_t at, int) const {
        return {Vec<V>::load(d + at), Vec<V>::load(g + at), Vec<V>::load(gt + at)};
    }
    template <int V> __device__ void apply(const In<V>& in, size_t, int p, int, double (&acc)[4]) const {
        const double a = alpha[p];
        for (int i = 0; i < V; ++i) {
            acc[0] = fmax(acc[0], fabs(a * in.d.v[i]));
            acc[1] += in.gt.v[i] * in.gt.v[i];
            acc[2] += in.gt.v[i] * (in.g.v[i] - in.gt.v[i]);
            acc[3] += fabs(in.gt.v[i]);
        }
    }
};

// After the accept / reject decision.  Mode bits:
//   1  the problem moves to the trial point: x <- x_try, g <- g_try (optimize.py:188-195), then
//      the new direction from the new g
//   2  restart: d <- -g (:216-218) instead of d <- gamma d - g (:221-222)
//   4  the new direction from g as it stands: for a caller that has exchanged its (x, g) and
//      (x_try, g_try) arrays instead of copying, so that x and g already hold the accepted point
//   8  copy only: x <- x_try, g <- g_try, d untouched, results zero (rows that the exchange
//      must not move: rejected and finished problems)
// Returns {d.g, d.d, g.g} of the new pair, which is what the next iteration starts from.
struct Update {
    static constexpr int K = 3;
    static constexpr unsigned MAXMASK = 0;
    const double *xt, *gt, *gamma;
    double *x, *g, *d;
    template <int V> struct In { Vec<V> xt, g, d; };
    template <int V> __device__ In<V> load(size_t at, int md) const {
        In<V> in;
        in.g = Vec<V>::load(((md & 9) ? gt : g) + at);
        if (md & 9) in.xt = Vec<V>::load(xt + at);
        if (!(md & 10)) in.d = Vec<V>::load(d + at);
        return in;
    }
    template <int V> __device__ void apply(const In<V>& in, size_t at, int p, int md, double (&acc)[4]) const {
        if (md & 9) {
            in.xt.store(x + at);
            in.g.store(g + at);
        }
        if (md & 8) return;
        const double gm = gamma[p];
        Vec<V> dn;
        for (int i = 0; i < V; ++i) {
            dn.v[i] = (md & 2) ? -in.g.v[i] : gm * in.d.v[i] - in.g.v[i];
            acc[0] += dn.v[i] * in.g.v[i];
            acc[1] += dn.v[i] * dn.v[i];
            acc[2] += in.g.v[i] * in.g.v[i];
        }
        dn.store(d + at);
    }
};

template <int V, class Op>
__global__ void __launch_bounds__(NT) scg_pass_kernel(Op op, Geom geo, const int* __restrict__ mode, int nch,
                                                      double* __restrict__ work) {
    const int ch = blockIdx.x, p = blockIdx.y, z = blockIdx.z;
    const int md = mode ? mode[p] : 1;
    if (md == 0) return;                                   // the whole block leaves together
    const long long len = (z ? geo.nb : geo.nA) / V;       // units in this segment
    const size_t base = seg_offset(geo, p, z);
    const long long per = (len + nch - 1) / nch;
    const long long lo = (long long)ch * per;
    const long long hi = lo + per < len ? lo + per : len;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long u0 = lo + threadIdx.x; u0 < hi; u0 += (long long)NT * U) {
        typename Op::template In<V> in[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + (long long)j * NT;
            if (u < hi) in[j] = op.template load<V>(base + (size_t)u * V, md);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long u = u0 + (long long)j * NT;
            if (u < hi) op.template apply<V>(in[j], base + (size_t)u * V, p, md, acc);
        }
    }
    if constexpr (Op::K > 0) {
        __shared__ double red[NT / 64][4];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < Op::K; ++k) {
            double v = acc[k];
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(v, off, 64);
                v = ((Op::MAXMASK >> k) & 1u) ? fmax(v, o) : v + o;
            }
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < Op::K) {
            const int k = threadIdx.x;
            double v = red[0][k];
            for (int w = 1; w < NT / 64; ++w) v = ((Op::MAXMASK >> k) & 1u) ? fmax(v, red[w][k]) : v + red[w][k];
            work[(((size_t)p * 2 + z) * nch + ch) * 4 + k] = v;
        }
    }
}

// out[p][k] = chunks of problem p combined in index order (0 for a problem with mode 0).
__global__ void scg_finish_kernel(int P, int nch, int K, unsigned maxmask, const int* __restrict__ mode,
                                  const double* __restrict__ work, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * 4) return;
    const int p = i >> 2, k = i & 3;
    double v = 0.0;
    if (k < K && (!mode || mode[p] != 0)) {
        const double* w = work + (size_t)p * 2 * nch * 4 + k;
        const bool is_max = (maxmask >> k) & 1u;
        for (int c = 0; c < 2 * nch; ++c) v = is_max ? fmax(v, w[(size_t)c * 4]) : v + w[(size_t)c * 4];
    }
    out[i] = v;
}

int check_geom(const char* who, int P, long long nA, long long nb, int nch, bool reduces, const void* work, const void* out) {
    VGPA_REQUIRE(P >= 0 && nA >= 1 && nb >= 1, "%s: bad sizes P=%d nA=%lld nb=%lld", who, P, nA, nb);
    VGPA_REQUIRE(P <= 65535, "%s: P=%d too large (at most 65535 problems per call)", who, P);
    if (reduces) {
        VGPA_REQUIRE(nch >= 1 && nch <= 1024, "%s: nch=%d must be in [1, 1024]", who, nch);
        VGPA_REQUIRE(work && out, "%s: null work / out pointer", who);
    }
    return VGPA_OK;
}

template <class Op>
int run_pass(const char* who, const Op& op, int P, long long nA, long long nb, const int* mode, int nch, double* work,
             double* out, void* stream) {
    if (P == 0) return VGPA_OK;
    hipStream_t st = vgpa::as_stream(stream);
    const Geom geo{P, nA, nb};
    if (Op::K == 0) {                                      // no reduction: enough chunks to fill the device
        const long long longest = nA > nb ? nA : nb;
        nch = (int)((longest + (long long)NT * U * 8 - 1) / ((long long)NT * U * 8));
        nch = nch < 1 ? 1 : (nch > 1024 ? 1024 : nch);
    }
    const dim3 grid((unsigned)nch, (unsigned)P, 2);
    if (nA % 2 == 0 && nb % 2 == 0)
        hipLaunchKernelGGL((scg_pass_kernel<2, Op>), grid, dim3(NT), 0, st, op, geo, mode, nch, work);
    else
        hipLaunchKernelGGL((scg_pass_kernel<1, Op>), grid, dim3(NT), 0, st, op, geo, mode, nch, work);
    if (Op::K > 0)
        hipLaunchKernelGGL(scg_finish_kernel, dim3((unsigned)((P * 4 + 255) / 256)), dim3(256), 0, st, P, nch, Op::K,
                           Op::MAXMASK, mode, work, out);
    return vgpa::check_launch(who);
}

}  // namespace

extern "C" int vgpa_scg_dots(int P, long long nA, long long nb, const double* d, const double* g, const int* mode,
                             int nch, double* work, double* out, void* stream) {
    if (int rc = check_geom("vgpa_scg_dots", P, nA, nb, nch, true, work, out)) return rc;
    VGPA_REQUIRE(d && g, "vgpa_scg_dots: null pointer argument");
    return run_pass("vgpa_scg_dots", Dots3{d, g}, P, nA, nb, mode, nch, work, out, stream);
}

extern "C" int vgpa_scg_step(int P, long long nA, long long nb, const double* x, double* d, const double* g,
                             const double* coef, const int* mode, double* y, void* stream) {
    if (int rc = check_geom("vgpa_scg_step", P, nA, nb, 1, false, nullptr, nullptr)) return rc;
    VGPA_REQUIRE(x && d && g && coef && mode && y, "vgpa_scg_step: null pointer argument");
    return run_pass("vgpa_scg_step", Step{x, g, coef, d, y}, P, nA, nb, mode, 1, nullptr, nullptr, stream);
}

extern "C" int vgpa_scg_curvature(int P, long long nA, long long nb, const double* d, const double* g_probe,
                                  const double* g, const int* mode, int nch, double* work, double* out, void* stream) {
    if (int rc = check_geom("vgpa_scg_curvature", P, nA, nb, nch, true, work, out)) return rc;
    VGPA_REQUIRE(d && g_probe && g, "vgpa_scg_curvature: null pointer argument");
    return run_pass("vgpa_scg_curvature", Curvature{d, g_probe, g}, P, nA, nb, mode, nch, work, out, stream);
}

extern "C" int vgpa_scg_trial_stats(int P, long long nA, long long nb, const double* d, const double* g,
                                    const double* g_try, const double* alpha, const int* mode, int nch, double* work,
                                    double* out, void* stream) {
    if (int rc = check_geom("vgpa_scg_trial_stats", P, nA, nb, nch, true, work, out)) return rc;
    VGPA_REQUIRE(d && g && g_try && alpha, "vgpa_scg_trial_stats: null pointer argument");
    return run_pass("vgpa_scg_trial_stats", TrialStats{d, g, g_try, alpha}, P, nA, nb, mode, nch, work, out, stream);
}

extern "C" int vgpa_scg_update(int P, long long nA, long long nb, double* x, const double* x_try, double* g,
                               const double* g_try, double* d, const double* gamma, const int* mode, int nch,
                               double* work, double* out, void* stream) {
    if (int rc = check_geom("vgpa_scg_update", P, nA, nb, nch, true, work, out)) return rc;
    VGPA_REQUIRE(x && x_try && g && g_try && d && gamma && mode, "vgpa_scg_update: null pointer argument");
    return run_pass("vgpa_scg_update", Update{x_try, g_try, gamma, x, g, d}, P, nA, nb, mode, nch, work, out, stream);
}
