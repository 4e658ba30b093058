// Forward moment scans m(t), S(t) and backward multiplier scans lam(t), Psi(t)
// for gfx950.
//
// For D > 1 the recurrences are sequential in time, so parallelism comes from the batch of
// independent problems (one workgroup per problem; D = 2, 3: sixteen problems per wavefront,
// small_d.h) and from the D x D matrix products inside a step.  For D == 1 a step is an affine
// map of a scalar and the scan is a parallel prefix computation (scan_1d_prefix.h).
//
// n-D layout inside a workgroup: every thread owns one 2 x 4 tile of the D x D
// state (S or Psi) and keeps it, and its Runge-Kutta increments, in registers for
// the whole scan.  Only the stage input X (what the other threads need for the
// matrix product) goes through LDS, in two alternating buffers so that one
// barrier per stage is enough.  A(t) is staged through LDS and prefetched two
// steps ahead into registers.
//
// Two orthogonal specialisations:
//   DT   compile-time dimension (40, 16) with an even leading dimension, 16-byte
//        LDS reads and unrolled product loops; DT = 0 is the run-time-D version.
//   SYM  the caller guarantees bitwise symmetric inputs (S0, Sigma; or dEsde_dS and
//        the jumps).  Then -A S - S A' = -(M + M') with M = A S, and
//        Psi A + A' Psi = Q + Q' with Q = Psi A: one product per stage instead of
//        two, the transpose going through LDS.  Because the same products are
//        summed in the same order, the result is bit-for-bit the general one
//        (not at D = 40: scan_mfma.h and scan_mfma_stages.h do that product on the
//        matrix cores, in another summation order).
#include <type_traits>

#include "common.h"
#include "sym_rows.h"

namespace {

constexpr int TI = 2;    // tile rows per thread
constexpr int TJ = 4;    // tile cols per thread
constexpr int PRE = 8;   // max A elements a thread stages per step (D*D / nthreads <= 8)

template <int DT_>
struct NdGeom {
    static constexpr int DT = DT_;
    int Dr, LDr;          // run-time dimension / leading dimension (DT == 0)
    int nthreads;
    int i0, j0;           // tile origin
    bool has_tile;
    int vi;               // row of the vector state owned by this thread, or -1
    __device__ __forceinline__ int D() const { return DT > 0 ? DT : Dr; }
    __device__ __forceinline__ int LD() const { return DT > 0 ? DT + 2 : LDr; }
    __device__ __forceinline__ int MS() const { return D() * LD(); }
};

template <int DT>
__device__ inline NdGeom<DT> make_geom(int D) {
    NdGeom<DT> g;
    g.Dr = D;
    g.LDr = D + 1;
    g.nthreads = blockDim.x;
    const int ntj = (g.D() + TJ - 1) / TJ;
    const int nti = (g.D() + TI - 1) / TI;
    const int tile = threadIdx.x;
    g.has_tile = tile < nti * ntj;
    g.i0 = (tile / ntj) * TI;
    g.j0 = (tile % ntj) * TJ;
    const int v = g.nthreads - 1 - (int)threadIdx.x;   // vector rows live on the last D threads
    g.vi = (v < g.D()) ? v : -1;
    return g;
}

__device__ inline int clampi(int x, int hi) { return x < hi ? x : hi; }

template <class G>
__device__ inline void load_matrix_lds(const G& g, const double* __restrict__ src, double* dst) {
    const int D = g.D(), DD = D * D;
    for (int e = threadIdx.x; e < DD; e += g.nthreads) dst[(e / D) * g.LD() + (e % D)] = src[e];
}

// For the compile-time sizes the block has exactly one thread per tile and D*D is
// a multiple of 2 * nthreads, so the staging loads, tile loads and tile stores
// below carry no predicates: the compiler can then count outstanding memory
// operations and wait only for the oldest ones (a predicated load forces a full
// drain of the prefetch queue at every use).
template <class G>
__device__ inline void prefetch_matrix(const G& g, const double* __restrict__ src, double pre[PRE]) {
    if constexpr (G::DT > 0) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
        for (int q = 0; q < PRE / 2; ++q) {
            const double2 v = s2[threadIdx.x + q * (G::DT * G::DT / PRE)];
            pre[2 * q] = v.x;
            pre[2 * q + 1] = v.y;
        }
    } else {
        const int DD = g.D() * g.D();
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = threadIdx.x + q * g.nthreads;
            pre[q] = (e < DD) ? src[e] : 0.0;
        }
    }
}

// Write the prefetched matrix into `dst` and, when `mid` is non-null, the midpoint
// 0.5*(other + new) into `mid` (each element is handled by the thread that staged it).
template <class G>
__device__ inline void commit_matrix(const G& g, const double pre[PRE], double* dst,
                                     const double* other, double* mid) {
    if constexpr (G::DT > 0) {
#pragma unroll
        for (int q = 0; q < PRE / 2; ++q) {
            const int e = 2 * (threadIdx.x + q * (G::DT * G::DT / PRE));
            const int o = (e / G::DT) * g.LD() + (e % G::DT);
            *reinterpret_cast<double2*>(dst + o) = make_double2(pre[2 * q], pre[2 * q + 1]);
            if (mid) {
                const double2 ov = *reinterpret_cast<const double2*>(other + o);
                *reinterpret_cast<double2*>(mid + o) =
                    make_double2(0.5 * (ov.x + pre[2 * q]), 0.5 * (ov.y + pre[2 * q + 1]));
            }
        }
    } else {
        const int D = g.D(), DD = D * D;
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = threadIdx.x + q * g.nthreads;
            if (e < DD) {
                const int o = (e / D) * g.LD() + (e % D);
                dst[o] = pre[q];
                if (mid) mid[o] = 0.5 * (other[o] + pre[q]);
            }
        }
    }
}

template <class G>
__device__ inline void load_tile(const G& g, const double* __restrict__ src, double t[TI][TJ]) {
    if constexpr (G::DT > 0) {
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            const double2* r = reinterpret_cast<const double2*>(src + (g.i0 + a) * G::DT + g.j0);
            const double2 v0 = r[0], v1 = r[1];
            t[a][0] = v0.x; t[a][1] = v0.y; t[a][2] = v1.x; t[a][3] = v1.y;
        }
    } else {
        const int D = g.D();
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                const int i = g.i0 + a, j = g.j0 + c;
                t[a][c] = (g.has_tile && i < D && j < D) ? src[i * D + j] : 0.0;
            }
    }
}

template <class G>
__device__ inline void store_tile_global(const G& g, double* __restrict__ dst, const double t[TI][TJ]) {
    if constexpr (G::DT > 0) {
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            double2* r = reinterpret_cast<double2*>(dst + (g.i0 + a) * G::DT + g.j0);
            r[0] = make_double2(t[a][0], t[a][1]);
            r[1] = make_double2(t[a][2], t[a][3]);
        }
    } else {
        if (!g.has_tile) return;
        const int D = g.D();
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                const int i = g.i0 + a, j = g.j0 + c;
                if (i < D && j < D) dst[i * D + j] = t[a][c];
            }
    }
}

template <class G>
__device__ inline void store_tile_lds(const G& g, double* dst, const double t[TI][TJ]) {
    if constexpr (G::DT > 0) {
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            double2* r = reinterpret_cast<double2*>(dst + (g.i0 + a) * g.LD() + g.j0);
            r[0] = make_double2(t[a][0], t[a][1]);
            r[1] = make_double2(t[a][2], t[a][3]);
        }
    } else {
        if (!g.has_tile) return;
        const int D = g.D();
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                const int i = g.i0 + a, j = g.j0 + c;
                if (i < D && j < D) dst[i * g.LD() + j] = t[a][c];
            }
    }
}

// acc[a][c] = sum_q Lt[i0+a][q] * Rt[q][j0+c]   ("row x column" product)
template <class G>
__device__ inline void prod_rc(const G& g, const double* Lt, const double* Rt, double acc[TI][TJ]) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = 0.0;
    if (G::DT == 0 && !g.has_tile) return;
    const int LD = g.LD();
    if constexpr (G::DT > 0) {
        const double* l0 = Lt + g.i0 * LD;
        const double* l1 = l0 + LD;
        const double* r = Rt + g.j0;
#pragma unroll 5
        for (int q = 0; q < G::DT; q += 2) {
            const double2 a0 = *reinterpret_cast<const double2*>(l0 + q);
            const double2 a1 = *reinterpret_cast<const double2*>(l1 + q);
            const double2 x0 = *reinterpret_cast<const double2*>(r + q * LD);
            const double2 x1 = *reinterpret_cast<const double2*>(r + q * LD + 2);
            const double2 y0 = *reinterpret_cast<const double2*>(r + (q + 1) * LD);
            const double2 y1 = *reinterpret_cast<const double2*>(r + (q + 1) * LD + 2);
            const double xs[TJ] = {x0.x, x0.y, x1.x, x1.y}, ys[TJ] = {y0.x, y0.y, y1.x, y1.y};
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                acc[0][c] = fma(a0.x, xs[c], acc[0][c]);
                acc[1][c] = fma(a1.x, xs[c], acc[1][c]);
            }
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                acc[0][c] = fma(a0.y, ys[c], acc[0][c]);
                acc[1][c] = fma(a1.y, ys[c], acc[1][c]);
            }
        }
    } else {
        const int D = g.D();
        int ri[TI], cj[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) ri[a] = clampi(g.i0 + a, D - 1) * LD;
#pragma unroll
        for (int c = 0; c < TJ; ++c) cj[c] = clampi(g.j0 + c, D - 1);
        for (int q = 0; q < D; ++q) {
            double lv[TI], rv[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) lv[a] = Lt[ri[a] + q];
#pragma unroll
            for (int c = 0; c < TJ; ++c) rv[c] = Rt[q * LD + cj[c]];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int c = 0; c < TJ; ++c) acc[a][c] = fma(lv[a], rv[c], acc[a][c]);
        }
    }
}

// acc[a][c] = sum_q Lt[i0+a][q] * Rt[j0+c][q]   ("row x row": L R')
template <class G>
__device__ inline void prod_rr(const G& g, const double* Lt, const double* Rt, double acc[TI][TJ]) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = 0.0;
    if (!g.has_tile) return;
    const int LD = g.LD(), D = g.D();
    int ri[TI], rj[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) ri[a] = clampi(g.i0 + a, D - 1) * LD;
#pragma unroll
    for (int c = 0; c < TJ; ++c) rj[c] = clampi(g.j0 + c, D - 1) * LD;
    if constexpr (G::DT > 0) {
#pragma unroll 5
        for (int q = 0; q < G::DT; q += 2) {
            double2 lv[TI], rv[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) lv[a] = *reinterpret_cast<const double2*>(Lt + ri[a] + q);
#pragma unroll
            for (int c = 0; c < TJ; ++c) rv[c] = *reinterpret_cast<const double2*>(Rt + rj[c] + q);
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int c = 0; c < TJ; ++c) acc[a][c] = fma(lv[a].x, rv[c].x, acc[a][c]);
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int c = 0; c < TJ; ++c) acc[a][c] = fma(lv[a].y, rv[c].y, acc[a][c]);
        }
    } else {
        for (int q = 0; q < D; ++q) {
            double lv[TI], rv[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) lv[a] = Lt[ri[a] + q];
#pragma unroll
            for (int c = 0; c < TJ; ++c) rv[c] = Rt[rj[c] + q];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int c = 0; c < TJ; ++c) acc[a][c] = fma(lv[a], rv[c], acc[a][c]);
        }
    }
}

// acc[a][c] = sum_q Lt[q][i0+a] * Rt[q][j0+c]   ("column x column": L' R)
template <class G>
__device__ inline void prod_cc(const G& g, const double* Lt, const double* Rt, double acc[TI][TJ]) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = 0.0;
    if (!g.has_tile) return;
    const int LD = g.LD(), D = g.D();
    int ci[TI], cj[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) ci[a] = clampi(g.i0 + a, D - 1);
#pragma unroll
    for (int c = 0; c < TJ; ++c) cj[c] = clampi(g.j0 + c, D - 1);
#pragma unroll 4
    for (int q = 0; q < D; ++q) {
        double lv[TI], rv[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) lv[a] = Lt[q * LD + ci[a]];
#pragma unroll
        for (int c = 0; c < TJ; ++c) rv[c] = Rt[q * LD + cj[c]];
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) acc[a][c] = fma(lv[a], rv[c], acc[a][c]);
    }
}

// Transposed tile read: t[a][c] = M[j0+c][i0+a]
template <class G>
__device__ inline void load_tile_T(const G& g, const double* M, double t[TI][TJ]) {
    const int D = g.D();
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            const int i = g.i0 + a, j = g.j0 + c;
            t[a][c] = (g.has_tile && i < D && j < D) ? M[j * g.LD() + i] : 0.0;
        }
}

#define TILE_FOR(a, c) \
    _Pragma("unroll") for (int a = 0; a < TI; ++a) _Pragma("unroll") for (int c = 0; c < TJ; ++c)

// k = Sig - Am*X - X*Am'   (core/fwd_ode.py:45).  Contains a barrier when SYM.
template <bool SYM, class G>
__device__ inline void fwd_rhs_tile(const G& g, const double* Am, const double* X, double* Mt,
                                    const double sig[TI][TJ], double k[TI][TJ]) {
    double s1[TI][TJ], s2[TI][TJ];
    prod_rc(g, Am, X, s1);                 // A X
    if constexpr (SYM) {
        store_tile_lds(g, Mt, s1);
        __syncthreads();
        load_tile_T(g, Mt, s2);            // (A X)' = X A' for symmetric X
    } else {
        prod_rr(g, X, Am, s2);             // X A'
    }
    TILE_FOR(a, c) k[a][c] = -s1[a][c] - s2[a][c] + sig[a][c];
}

// k = -G + X*Am + Am'*X   (core/bwd_ode.py:43).  Contains a barrier when SYM.
template <bool SYM, class G>
__device__ inline void bwd_rhs_tile(const G& g, const double* Am, const double* X, double* Mt,
                                    const double Gt[TI][TJ], double k[TI][TJ]) {
    double s1[TI][TJ], s2[TI][TJ];
    prod_rc(g, X, Am, s1);                 // X A
    if constexpr (SYM) {
        store_tile_lds(g, Mt, s1);
        __syncthreads();
        load_tile_T(g, Mt, s2);            // (X A)' = A' X for symmetric X
    } else {
        prod_cc(g, Am, X, s2);             // A' X
    }
    // (s1 + s2) first: that sum is symmetric in (i, j) bit for bit, which keeps Psi exactly
    // symmetric whenever G and the jumps are.
    TILE_FOR(a, c) k[a][c] = -Gt[a][c] + (s1[a][c] + s2[a][c]);
}

// (Am x)[vi]
template <class G>
__device__ inline double matvec_row(const G& g, const double* Am, const double* x) {
    double acc0 = 0.0, acc1 = 0.0;
    const double* row = Am + g.vi * g.LD();
    const int D = g.D();
    int q = 0;
#pragma unroll 4
    for (; q + 1 < D; q += 2) {
        acc0 = fma(row[q], x[q], acc0);
        acc1 = fma(row[q + 1], x[q + 1], acc1);
    }
    if (q < D) acc0 = fma(row[q], x[q], acc0);
    return acc0 + acc1;
}

// LDS carve-up shared by both directions: two A buffers, two X buffers, then the
// midpoint A (RK2 / RK4 only), the transpose buffer (SYM only) and five vectors.
// Buffers are addressed as offsets from the one dynamic LDS array: a table of
// pointers selected at run time would hide the address space from the compiler
// and turn every LDS access into a flat access.
template <bool MID, bool SYM>
struct LdsMap {
    int MS;
    __device__ explicit LdsMap(int ms) : MS(ms) {}
    __device__ __forceinline__ int A(int k) const { return k * MS; }
    __device__ __forceinline__ int X(int k) const { return (2 + k) * MS; }
    __device__ __forceinline__ int Amid() const { return 4 * MS; }
    __device__ __forceinline__ int Mt() const { return (4 + (MID ? 1 : 0)) * MS; }
    __device__ __forceinline__ int vec() const { return (4 + (MID ? 1 : 0) + (SYM ? 1 : 0)) * MS; }
};

// ---------------------------------------------------------------------------
// forward scan, D > 1: one workgroup per problem.
// core/fwd_ode.py: euler_nD :87, heun_nD :181, rk2_nD :311, rk4_nD :426.
// ---------------------------------------------------------------------------
template <int METHOD, int DT, bool SYM>
__global__ void __launch_bounds__(DT == 16 ? 64 : 512)
fwd_scan_nd_kernel(int N, int Drt, const double* __restrict__ A, const double* __restrict__ b,
                   const double* __restrict__ m0, const double* __restrict__ S0,
                   const double* __restrict__ Sigma, int sigma_stride, double dt,
                   double* __restrict__ m_out, double* __restrict__ S_out) {
    extern __shared__ double lds[];
    using G = NdGeom<DT>;
    const G g = make_geom<DT>(Drt);
    const int D = g.D();
    constexpr bool MID = (METHOD == VGPA_RK2 || METHOD == VGPA_RK4);
    const LdsMap<MID, SYM> L(g.MS());
#define ABUF(k) (lds + L.A(k))
#define XBUF(k) (lds + L.X(k))
#define AMID (lds + L.Amid())
#define MTB (lds + L.Mt())
#define BBUF(k) (lds + L.vec() + (k) * D)
#define BMID (lds + L.vec() + 2 * D)
#define XVEC(k) (lds + L.vec() + (3 + (k)) * D)

    const int p = blockIdx.x;
    const size_t DD = (size_t)D * D;
    const double* Ap = A + (size_t)p * N * DD;
    const double* bp = b + (size_t)p * N * D;
    double* mo = m_out + (size_t)p * N * D;
    double* So = S_out + (size_t)p * N * DD;
    const double h = 0.5 * dt;

    double sig[TI][TJ], s[TI][TJ];
    load_tile(g, Sigma + (size_t)p * sigma_stride, sig);
    load_tile(g, S0 + (size_t)p * DD, s);
    double mv = (g.vi >= 0) ? m0[(size_t)p * D + g.vi] : 0.0;

    load_matrix_lds(g, Ap, ABUF(0));
    store_tile_lds(g, XBUF(0), s);
    store_tile_global(g, So, s);
    if (g.vi >= 0) {
        BBUF(0)[g.vi] = bp[g.vi];
        XVEC(0)[g.vi] = mv;
        mo[g.vi] = mv;
    }
    // Register prefetch, two steps deep: preA holds A[t+1], preB holds A[t+2].
    double preA[PRE], preB[PRE];
    const int vsafe = g.vi >= 0 ? g.vi : 0;
    const int r1 = N > 1 ? 1 : 0, r2 = N > 2 ? 2 : N - 1;
    prefetch_matrix(g, Ap + (size_t)r1 * DD, preA);
    prefetch_matrix(g, Ap + (size_t)r2 * DD, preB);
    double bA = bp[(size_t)r1 * D + vsafe], bB = bp[(size_t)r2 * D + vsafe];
    int cur = 0;   // A buffer holding A[t]
    int xin = 0;   // X buffer holding S[t] / m[t]

    auto step = [&](int t, double (&pre)[PRE], double& bnext) {
        __syncthreads();   // previous step's LDS writes visible, previous reads retired
        const int nxt = cur ^ 1;
        commit_matrix(g, pre, ABUF(nxt), ABUF(cur), MID ? AMID : nullptr);
        if (g.vi >= 0) {
            BBUF(nxt)[g.vi] = bnext;
            if (MID) BMID[g.vi] = 0.5 * (BBUF(cur)[g.vi] + bnext);
        }
        {   // rows past the end are clamped: a redundant load instead of a branch
            const int tp = (t + 3 < N) ? t + 3 : N - 1;
            prefetch_matrix(g, Ap + (size_t)tp * DD, pre);
            bnext = bp[(size_t)tp * D + vsafe];
        }

        double k1[TI][TJ], snew[TI][TJ];
        double mnew = 0.0;
        fwd_rhs_tile<SYM>(g, ABUF(cur), XBUF(xin), MTB, sig, k1);
        double f1 = 0.0;
        if (g.vi >= 0) f1 = -matvec_row(g, ABUF(cur), XVEC(xin)) + BBUF(cur)[g.vi];

        if (METHOD == VGPA_EULER) {
            TILE_FOR(a, c) snew[a][c] = s[a][c] + k1[a][c] * dt;
            mnew = mv + f1 * dt;
            store_tile_lds(g, XBUF(xin ^ 1), snew);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = mnew;
            xin ^= 1;
        } else if (METHOD == VGPA_HEUN) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] + k1[a][c] * dt;
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = mv + f1 * dt;
            __syncthreads();
            fwd_rhs_tile<SYM>(g, ABUF(nxt), XBUF(xin ^ 1), MTB, sig, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] + h * (k1[a][c] + k2[a][c]);
            if (g.vi >= 0) {
                const double f2 = -matvec_row(g, ABUF(nxt), XVEC(xin ^ 1)) + BBUF(nxt)[g.vi];
                mnew = mv + h * (f1 + f2);
            }
            store_tile_lds(g, XBUF(xin), snew);
            if (g.vi >= 0) XVEC(xin)[g.vi] = mnew;
        } else if (METHOD == VGPA_RK2) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] + h * k1[a][c];
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = mv + h * f1;
            __syncthreads();
            fwd_rhs_tile<SYM>(g, AMID, XBUF(xin ^ 1), MTB, sig, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] + k2[a][c] * dt;
            if (g.vi >= 0) {
                const double f2 = -matvec_row(g, AMID, XVEC(xin ^ 1)) + BMID[g.vi];
                mnew = mv + f2 * dt;
            }
            store_tile_lds(g, XBUF(xin), snew);
            if (g.vi >= 0) XVEC(xin)[g.vi] = mnew;
        } else {   // RK4
            double L1[TI][TJ], L2[TI][TJ], L3[TI][TJ], L4[TI][TJ], tmp[TI][TJ];
            double K1 = f1 * dt, K2 = 0.0, K3 = 0.0, K4 = 0.0;
            TILE_FOR(a, c) { L1[a][c] = k1[a][c] * dt; tmp[a][c] = s[a][c] + 0.5 * L1[a][c]; }
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = mv + 0.5 * K1;
            __syncthreads();
            fwd_rhs_tile<SYM>(g, AMID, XBUF(xin ^ 1), MTB, sig, L2);
            TILE_FOR(a, c) { L2[a][c] *= dt; tmp[a][c] = s[a][c] + 0.5 * L2[a][c]; }
            if (g.vi >= 0) K2 = (-matvec_row(g, AMID, XVEC(xin ^ 1)) + BMID[g.vi]) * dt;
            store_tile_lds(g, XBUF(xin), tmp);
            if (g.vi >= 0) XVEC(xin)[g.vi] = mv + 0.5 * K2;
            __syncthreads();
            fwd_rhs_tile<SYM>(g, AMID, XBUF(xin), MTB, sig, L3);
            TILE_FOR(a, c) { L3[a][c] *= dt; tmp[a][c] = s[a][c] + L3[a][c]; }
            if (g.vi >= 0) K3 = (-matvec_row(g, AMID, XVEC(xin)) + BMID[g.vi]) * dt;
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = mv + K3;
            __syncthreads();
            fwd_rhs_tile<SYM>(g, ABUF(nxt), XBUF(xin ^ 1), MTB, sig, L4);
            TILE_FOR(a, c) {
                L4[a][c] *= dt;
                snew[a][c] = s[a][c] + (L1[a][c] + 2.0 * (L2[a][c] + L3[a][c]) + L4[a][c]) / 6.0;
            }
            if (g.vi >= 0) {
                K4 = (-matvec_row(g, ABUF(nxt), XVEC(xin ^ 1)) + BBUF(nxt)[g.vi]) * dt;
                mnew = mv + (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            }
            store_tile_lds(g, XBUF(xin), snew);
            if (g.vi >= 0) XVEC(xin)[g.vi] = mnew;
        }
        TILE_FOR(a, c) s[a][c] = snew[a][c];
        store_tile_global(g, So + (size_t)(t + 1) * DD, s);
        if (g.vi >= 0) {
            mv = mnew;
            mo[(size_t)(t + 1) * D + g.vi] = mv;
        }
        cur = nxt;
    };

    for (int t = 0; t + 1 < N; t += 2) {
        step(t, preA, bA);
        if (t + 2 < N) step(t + 1, preB, bB);
    }
}

// ---------------------------------------------------------------------------
// backward scan, D > 1: one workgroup per problem.
// core/bwd_ode.py: euler_nD :86, heun_nD :177, rk2_nD :305, rk4_nD :427.
// ---------------------------------------------------------------------------
template <int METHOD, int DT, bool SYM>
__global__ void __launch_bounds__(DT == 16 ? 64 : 512)
bwd_scan_nd_kernel(int N, int Drt, const double* __restrict__ A, const double* __restrict__ dEdm,
                   const double* __restrict__ dEdS, const double* __restrict__ jm,
                   const double* __restrict__ jS_dense, const int* __restrict__ obs_flag,
                   const double* __restrict__ jumpS, int jumpS_stride, double dt,
                   const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                   double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    extern __shared__ double lds[];
    using G = NdGeom<DT>;
    const G g = make_geom<DT>(Drt);
    const int D = g.D();
    constexpr bool MID = (METHOD == VGPA_RK2 || METHOD == VGPA_RK4);
    const LdsMap<MID, SYM> L(g.MS());

    const int p = blockIdx.x;
    const size_t DD = (size_t)D * D;
    const double* Ap = A + (size_t)p * N * DD;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * DD;
    const double* jmp = jm + (size_t)p * N * D;
    const double* jSd = jS_dense ? jS_dense + (size_t)p * N * DD : nullptr;
    double* lo = lam_out + (size_t)p * N * D;
    double* Po = Psi_out + (size_t)p * N * DD;
    const double h = 0.5 * dt;

    double jt[TI][TJ];
    if (!jSd && jumpS && obs_flag) load_tile(g, jumpS + (size_t)p * jumpS_stride, jt);
    else { TILE_FOR(a, c) jt[a][c] = 0.0; }

    double s[TI][TJ];   // Psi[t]: zero at the last row unless a terminal state is given
    if (Psi_last) load_tile(g, Psi_last + (size_t)p * DD, s);
    else { TILE_FOR(a, c) s[a][c] = 0.0; }
    double lv = (lam_last && g.vi >= 0) ? lam_last[(size_t)p * D + g.vi] : 0.0;    // lam[t][vi]
    store_tile_global(g, Po + (size_t)(N - 1) * DD, s);
    store_tile_lds(g, XBUF(0), s);
    if (g.vi >= 0) {
        lo[(size_t)(N - 1) * D + g.vi] = lv;
        XVEC(0)[g.vi] = lv;
    }
    load_matrix_lds(g, Ap + (size_t)(N - 1) * DD, ABUF(0));
    // Register prefetch two steps deep: set A holds row t-1, set B holds row t-2.
    double preA[PRE], preB[PRE], GA[TI][TJ], GB[TI][TJ];
    const int vsafe = g.vi >= 0 ? g.vi : 0;
    const int r1 = N > 1 ? N - 2 : 0, r2 = N > 2 ? N - 3 : 0;
    prefetch_matrix(g, Ap + (size_t)r1 * DD, preA);
    load_tile(g, gS + (size_t)r1 * DD, GA);
    prefetch_matrix(g, Ap + (size_t)r2 * DD, preB);
    load_tile(g, gS + (size_t)r2 * DD, GB);
    double gA = gm[(size_t)r1 * D + vsafe], gB = gm[(size_t)r2 * D + vsafe];
    double Gt[TI][TJ];
    load_tile(g, gS + (size_t)(N - 1) * DD, Gt);
    double gmt = (g.vi >= 0) ? gm[(size_t)(N - 1) * D + g.vi] : 0.0;
    int cur = 0, xin = 0;

    auto step = [&](int t, double (&pre)[PRE], double (&Gpre)[TI][TJ], double& gpre) {
        __syncthreads();
        const int prv = cur ^ 1;   // buffer for A[t-1]
        commit_matrix(g, pre, ABUF(prv), ABUF(cur), MID ? AMID : nullptr);
        double Gp[TI][TJ];
        TILE_FOR(a, c) Gp[a][c] = Gpre[a][c];
        const double gmp = gpre;
        {   // rows before the start are clamped: a redundant load instead of a branch
            const int tp = (t >= 3) ? t - 3 : 0;
            prefetch_matrix(g, Ap + (size_t)tp * DD, pre);
            load_tile(g, gS + (size_t)tp * DD, Gpre);
            gpre = gm[(size_t)tp * D + vsafe];
        }
        // observation jumps at t-1 (core/bwd_ode.py:124,127), fetched early
        double jd[TI][TJ];
        if (jSd) {
            load_tile(g, jSd + (size_t)(t - 1) * DD, jd);
        } else {
            const bool on = obs_flag && obs_flag[t - 1];
            TILE_FOR(a, c) jd[a][c] = on ? jt[a][c] : 0.0;
        }
        const double jmv = jmp[(size_t)(t - 1) * D + vsafe];

        double k1[TI][TJ], snew[TI][TJ];
        double lnew = 0.0;
        bwd_rhs_tile<SYM>(g, ABUF(cur), XBUF(xin), MTB, Gt, k1);
        double f1 = 0.0;
        if (g.vi >= 0) f1 = -gmt + matvec_row(g, ABUF(cur), XVEC(xin));

        if (METHOD == VGPA_EULER) {
            TILE_FOR(a, c) snew[a][c] = s[a][c] - k1[a][c] * dt;
            lnew = lv - f1 * dt;
        } else if (METHOD == VGPA_HEUN) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] - k1[a][c] * dt;
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = lv - f1 * dt;
            __syncthreads();
            bwd_rhs_tile<SYM>(g, ABUF(prv), XBUF(xin ^ 1), MTB, Gp, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] - h * (k1[a][c] + k2[a][c]);
            if (g.vi >= 0) {
                const double f2 = -gmp + matvec_row(g, ABUF(prv), XVEC(xin ^ 1));
                lnew = lv - h * (f1 + f2);
            }
            xin ^= 1;   // now names the buffer read by the last stage
        } else if (METHOD == VGPA_RK2) {
            double tmp[TI][TJ], k2[TI][TJ], Gk[TI][TJ];
            TILE_FOR(a, c) { tmp[a][c] = s[a][c] - h * k1[a][c]; Gk[a][c] = 0.5 * (Gp[a][c] + Gt[a][c]); }
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = lv - h * f1;
            __syncthreads();
            bwd_rhs_tile<SYM>(g, AMID, XBUF(xin ^ 1), MTB, Gk, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] - k2[a][c] * dt;
            if (g.vi >= 0) {
                const double f2 = -0.5 * (gmp + gmt) + matvec_row(g, AMID, XVEC(xin ^ 1));
                lnew = lv - f2 * dt;
            }
            xin ^= 1;
        } else {   // RK4
            double L1[TI][TJ], L2[TI][TJ], L3[TI][TJ], L4[TI][TJ], tmp[TI][TJ], Gk[TI][TJ];
            const double gmk = 0.5 * (gmp + gmt);
            double K1 = f1 * dt, K2 = 0.0, K3 = 0.0, K4 = 0.0;
            TILE_FOR(a, c) {
                L1[a][c] = k1[a][c] * dt;
                tmp[a][c] = s[a][c] - 0.5 * L1[a][c];
                Gk[a][c] = 0.5 * (Gp[a][c] + Gt[a][c]);
            }
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = lv - 0.5 * K1;
            __syncthreads();
            bwd_rhs_tile<SYM>(g, AMID, XBUF(xin ^ 1), MTB, Gk, L2);
            TILE_FOR(a, c) { L2[a][c] *= dt; tmp[a][c] = s[a][c] - 0.5 * L2[a][c]; }
            if (g.vi >= 0) K2 = (-gmk + matvec_row(g, AMID, XVEC(xin ^ 1))) * dt;
            store_tile_lds(g, XBUF(xin), tmp);
            if (g.vi >= 0) XVEC(xin)[g.vi] = lv - 0.5 * K2;
            __syncthreads();
            bwd_rhs_tile<SYM>(g, AMID, XBUF(xin), MTB, Gk, L3);
            TILE_FOR(a, c) { L3[a][c] *= dt; tmp[a][c] = s[a][c] - L3[a][c]; }
            if (g.vi >= 0) K3 = (-gmk + matvec_row(g, AMID, XVEC(xin))) * dt;
            store_tile_lds(g, XBUF(xin ^ 1), tmp);
            if (g.vi >= 0) XVEC(xin ^ 1)[g.vi] = lv - K3;
            __syncthreads();
            bwd_rhs_tile<SYM>(g, ABUF(prv), XBUF(xin ^ 1), MTB, Gp, L4);
            TILE_FOR(a, c) {
                L4[a][c] *= dt;
                snew[a][c] = s[a][c] - (L1[a][c] + 2.0 * (L2[a][c] + L3[a][c]) + L4[a][c]) / 6.0;
            }
            if (g.vi >= 0) {
                K4 = (-gmp + matvec_row(g, ABUF(prv), XVEC(xin ^ 1))) * dt;
                lnew = lv - (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            }
            xin ^= 1;
        }
        TILE_FOR(a, c) snew[a][c] += jd[a][c];
        lnew += jmv;
        // `xin` names the buffer the last stage read; the new state goes to the other one.
        xin ^= 1;
        TILE_FOR(a, c) { s[a][c] = snew[a][c]; Gt[a][c] = Gp[a][c]; }
        store_tile_lds(g, XBUF(xin), s);
        store_tile_global(g, Po + (size_t)(t - 1) * DD, s);
        if (g.vi >= 0) {
            lv = lnew;
            gmt = gmp;
            XVEC(xin)[g.vi] = lv;
            lo[(size_t)(t - 1) * D + g.vi] = lv;
        }
        cur = prv;
    };

    for (int t = N - 1; t >= 1; t -= 2) {
        step(t, preA, GA, gA);
        if (t - 1 >= 1) step(t - 1, preB, GB, gB);
    }
}

#include "scan_1d_prefix.h"

int nd_threads(int D, bool fixed) {
    const int ntiles = ((D + TI - 1) / TI) * ((D + TJ - 1) / TJ);
    if (fixed) return ntiles;      // D*D/8 >= D threads, no idle lanes, no predicates
    int n = ntiles > D ? ntiles : D;
    n = vgpa::round_up(n, 64);
    return n < 64 ? 64 : n;
}

// LD = D + 2 for the compile-time sizes, D + 1 otherwise; four to six matrices, five vectors.
size_t nd_lds_bytes(int D, bool fixed, int method, bool sym) {
    const size_t LD = fixed ? D + 2 : D + 1;
    const size_t nmat = 4 + ((method == VGPA_RK2 || method == VGPA_RK4) ? 1 : 0) + (sym ? 1 : 0);
    return sizeof(double) * (nmat * D * LD + 5 * (size_t)D);
}

// LDS of one gfx950 compute unit: the most a workgroup can ask for.
constexpr size_t kLdsLimit = 160 * 1024;

// Whether the n-D kernels can hold their matrices for this size and method at all, and whether
// the one-product (symmetric) variant, which keeps one more matrix, still fits.
bool nd_fits(int D, int method, bool sym) {
    const bool fixed = (D == 40 || D == 16);
    return nd_lds_bytes(D, fixed, method, sym) <= kLdsLimit;
}

// Records a refused attribute change: the launch that follows would fail with a generic
// error, so the reason is kept for vgpa_last_error_string().
thread_local bool g_lds_refused = false;
template <typename K>
void allow_lds(K kernel, size_t bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        g_lds_refused = true;
    }
}

#include "scan_mfma.h"
#include "scan_mfma_stages.h"
#include "scan_fused.h"
#include "scan_fused_stages2.h"
#define VGPA_SMALL_D_SCANS
#include "small_d.h"
#include "small_d_chunks.h"

// The LDS-staged D = 2, 3 scans move 16 bytes per lane on the caller's arrays: every array they touch
// that way has to start on a 16-byte boundary (a contiguous view that begins at an odd row of a D = 3
// array is only 8-byte aligned); the per-lane kernels take over otherwise.
template <typename... Ptr>
bool all_16_byte_aligned(Ptr... p) {
    return (((reinterpret_cast<size_t>(p) & 15) == 0) && ...);
}

template <int METHOD, bool SYM>
bool launch_fwd_small(int P, int N, int D, const double* A, const double* b, const double* m0, const double* S0,
                      const double* Sigma, int sigma_stride, double dt, double* m_out, double* S_out, hipStream_t st) {
    if ((D == 2 || D == 3) && SmallChunks::use(P, N)) {
        // few problems: 16 (32 with symmetric state) chunks of the time axis per problem side by side
        if (D == 2) hipLaunchKernelGGL((fwd_scan_small_chunks_kernel<2, METHOD, SYM>), dim3(P), dim3(256), 0, st, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out);
        else hipLaunchKernelGGL((fwd_scan_small_chunks_kernel<3, METHOD, SYM>), dim3(P), dim3(256), 0, st, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out);
        return true;
    }
    // RK4 evaluates the moment equations four times per step out of registers: its time is the
    // arithmetic chain of one lane, which the register-ring kernel below keeps shorter.
    if ((D == 2 || D == 3) && ((long long)N * D) % 2 == 0 && METHOD != VGPA_RK4 &&
        all_16_byte_aligned(A, b, m_out, S_out)) {
        // 16-byte pieces through LDS, sixteen problems per wavefront
        const unsigned gl = (unsigned)((P + SmallLds::PB - 1) / SmallLds::PB);
        if constexpr (METHOD == VGPA_EULER) {
            // a row of the state per lane
            const size_t lds = sizeof(double) * SmallRows::fwd_lds_doubles(D);
            if (D == 2) allow_lds(fwd_scan_small_rows_kernel<2>, lds); else allow_lds(fwd_scan_small_rows_kernel<3>, lds);
            if (D == 2) hipLaunchKernelGGL(fwd_scan_small_rows_kernel<2>, dim3(gl), dim3(SmallRows::NT), lds, st, P, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out);
            else hipLaunchKernelGGL(fwd_scan_small_rows_kernel<3>, dim3(gl), dim3(SmallRows::NT), lds, st, P, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out);
            return true;
        }
#define VGPA_FL(DD_) hipLaunchKernelGGL((fwd_scan_small_lds_kernel<DD_, METHOD>), dim3(gl), dim3(64), 0, st, P, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out)
        if (D == 2) VGPA_FL(2); else VGPA_FL(3);
#undef VGPA_FL
        return true;
    }
    const unsigned grid = (unsigned)((P + 63) / 64);
#define VGPA_FS(DD_) hipLaunchKernelGGL((fwd_scan_small_kernel<DD_, METHOD>), dim3(grid), dim3(64), 0, st, P, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out)
    if (D == 2) VGPA_FS(2); else if (D == 3) VGPA_FS(3); else return false;
#undef VGPA_FS
    return true;
}

template <int METHOD, bool SYM>
bool launch_bwd_small(int P, int N, int D, const double* A, const double* dEdm, const double* dEdS, const double* jm,
                      const double* jS, const int* obs_flag, const double* jumpS, int jumpS_stride, double dt,
                      const double* lam_last, const double* Psi_last, double* lam_out, double* Psi_out, hipStream_t st) {
    if ((D == 2 || D == 3) && !jS && SmallChunks::use(P, N)) {
        if (D == 2) hipLaunchKernelGGL((bwd_scan_small_chunks_kernel<2, METHOD, SYM>), dim3(P), dim3(256), 0, st, N, A, dEdm, dEdS, jm, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
        else hipLaunchKernelGGL((bwd_scan_small_chunks_kernel<3, METHOD, SYM>), dim3(P), dim3(256), 0, st, N, A, dEdm, dEdS, jm, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
        return true;
    }
    if ((D == 2 || D == 3) && ((long long)N * D) % 2 == 0 && !jS &&
        all_16_byte_aligned(A, dEdm, dEdS, jm, lam_out, Psi_out)) {
        const unsigned gl = (unsigned)((P + SmallLds::PB - 1) / SmallLds::PB);
        if constexpr (METHOD == VGPA_EULER) {
            const size_t lds = sizeof(double) * SmallRows::bwd_lds_doubles(D);
            if (D == 2) allow_lds(bwd_scan_small_rows_kernel<2>, lds); else allow_lds(bwd_scan_small_rows_kernel<3>, lds);
            if (D == 2) hipLaunchKernelGGL(bwd_scan_small_rows_kernel<2>, dim3(gl), dim3(SmallRows::NT), lds, st, P, N, A, dEdm, dEdS, jm, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
            else hipLaunchKernelGGL(bwd_scan_small_rows_kernel<3>, dim3(gl), dim3(SmallRows::NT), lds, st, P, N, A, dEdm, dEdS, jm, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
            return true;
        }
#define VGPA_BL(DD_) hipLaunchKernelGGL((bwd_scan_small_lds_kernel<DD_, METHOD>), dim3(gl), dim3(64), 0, st, P, N, A, dEdm, dEdS, jm, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out)
        if (D == 2) VGPA_BL(2); else VGPA_BL(3);
#undef VGPA_BL
        return true;
    }
    const unsigned grid = (unsigned)((P + 63) / 64);
#define VGPA_BS(DD_) hipLaunchKernelGGL((bwd_scan_small_kernel<DD_, METHOD>), dim3(grid), dim3(64), 0, st, P, N, A, dEdm, dEdS, jm, jS, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out)
    if (D == 2) VGPA_BS(2); else if (D == 3) VGPA_BS(3); else return false;
#undef VGPA_BS
    return true;
}

template <int METHOD, int DT, bool SYM>
void launch_fwd(int P, int N, int D, const double* A, const double* b, const double* m0, const double* S0,
                const double* Sigma, int sigma_stride, double dt, double* m_out, double* S_out, hipStream_t st) {
    if constexpr (DT == 0) {
        if (launch_fwd_small<METHOD, SYM>(P, N, D, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out, st)) return;
    }
    if constexpr (METHOD == VGPA_EULER && DT == 40 && SYM) {
        const size_t lds = sizeof(double) * M40::LDS_DOUBLES;
        allow_lds(fwd_euler_sym_mfma40_kernel<false>, lds);
        hipLaunchKernelGGL(fwd_euler_sym_mfma40_kernel<false>, dim3(P), dim3(M40::NT), lds, st, N, A, b, m0, S0, Sigma,
                           sigma_stride, dt, m_out, S_out);
        return;
    }
    if constexpr (METHOD != VGPA_EULER && DT == 40 && SYM) {
        const size_t lds = sizeof(double) * M40S<METHOD>::LDS_DOUBLES;
        allow_lds(fwd_stages_sym_mfma40_kernel<METHOD>, lds);
        hipLaunchKernelGGL(fwd_stages_sym_mfma40_kernel<METHOD>, dim3(P), dim3(M40::NT), lds, st, N, A, b, m0, S0, Sigma,
                           sigma_stride, dt, m_out, S_out);
        return;
    }
    const size_t lds = nd_lds_bytes(D, DT > 0, METHOD, SYM);
    allow_lds(fwd_scan_nd_kernel<METHOD, DT, SYM>, lds);
    hipLaunchKernelGGL((fwd_scan_nd_kernel<METHOD, DT, SYM>), dim3(P), dim3(nd_threads(D, DT > 0)), lds, st, N, D, A, b, m0,
                       S0, Sigma, sigma_stride, dt, m_out, S_out);
}

template <int METHOD, int DT, bool SYM>
void launch_bwd(int P, int N, int D, const double* A, const double* dEdm, const double* dEdS, const double* jm,
                const double* jS, const int* obs_flag, const double* jumpS, int jumpS_stride, double dt,
                const double* lam_last, const double* Psi_last, double* lam_out, double* Psi_out, hipStream_t st) {
    if constexpr (DT == 0) {
        if (launch_bwd_small<METHOD, SYM>(P, N, D, A, dEdm, dEdS, jm, jS, obs_flag, jumpS, jumpS_stride, dt, lam_last,
                                     Psi_last, lam_out, Psi_out, st)) return;
    }
    if constexpr (METHOD == VGPA_EULER && DT == 40 && SYM) {
        if (!jS) {
            const size_t lds = sizeof(double) * (M40::LDS_DOUBLES + M40::MS);
            allow_lds(bwd_euler_sym_mfma40_kernel, lds);
            hipLaunchKernelGGL(bwd_euler_sym_mfma40_kernel, dim3(P), dim3(M40::NT), lds, st, N, A, dEdm, dEdS, jm,
                               obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
            return;
        }
    }
    if constexpr (METHOD != VGPA_EULER && DT == 40 && SYM) {
        if (!jS) {
            const size_t lds = sizeof(double) * M40S<METHOD>::LDS_DOUBLES;
            allow_lds(bwd_stages_sym_mfma40_kernel<METHOD>, lds);
            hipLaunchKernelGGL(bwd_stages_sym_mfma40_kernel<METHOD>, dim3(P), dim3(M40::NT), lds, st, N, A, dEdm, dEdS, jm,
                               obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
            return;
        }
    }
    const size_t lds = nd_lds_bytes(D, DT > 0, METHOD, SYM);
    allow_lds(bwd_scan_nd_kernel<METHOD, DT, SYM>, lds);
    hipLaunchKernelGGL((bwd_scan_nd_kernel<METHOD, DT, SYM>), dim3(P), dim3(nd_threads(D, DT > 0)), lds, st, N, D, A, dEdm,
                       dEdS, jm, jS, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out);
}

// Expand (method, D, symmetric) into the template instance.
#define VGPA_DISPATCH_SCAN(LAUNCH, ...)                                              \
    do {                                                                             \
        /* the symmetric variant keeps one more matrix in LDS: where that does not fit   \
           (RK2 / RK4 at D >= 58) the general kernel runs instead, with the same bits */   \
        const bool sym = (flags & VGPA_SCAN_SYMMETRIC) != 0 && nd_fits(D, method, true);  \
        const int dt_ = D == 40 ? 40 : (D == 16 ? 16 : 0);                           \
        VGPA_DISPATCH_M(LAUNCH, VGPA_EULER, dt_, sym, __VA_ARGS__);                  \
        VGPA_DISPATCH_M(LAUNCH, VGPA_HEUN, dt_, sym, __VA_ARGS__);                   \
        VGPA_DISPATCH_M(LAUNCH, VGPA_RK2, dt_, sym, __VA_ARGS__);                    \
        VGPA_DISPATCH_M(LAUNCH, VGPA_RK4, dt_, sym, __VA_ARGS__);                    \
    } while (0)
#define VGPA_DISPATCH_M(LAUNCH, M, dt_, sym, ...)                                    \
    if (method == M) {                                                               \
        if (dt_ == 40) { if (sym) LAUNCH<M, 40, true>(__VA_ARGS__); else LAUNCH<M, 40, false>(__VA_ARGS__); } \
        else if (dt_ == 16) { if (sym) LAUNCH<M, 16, true>(__VA_ARGS__); else LAUNCH<M, 16, false>(__VA_ARGS__); } \
        else { if (sym) LAUNCH<M, 0, true>(__VA_ARGS__); else LAUNCH<M, 0, false>(__VA_ARGS__); } \
    }

}  // namespace

extern "C" int vgpa_fwd_scan(int method, int P, int N, int D, const double* A, const double* b,
                             const double* m0, const double* S0, const double* Sigma, int sigma_stride,
                             double dt, int flags, double* m_out, double* S_out, void* stream) {
    VGPA_REQUIRE(method >= VGPA_EULER && method <= VGPA_RK4, "vgpa_fwd_scan: Unknown method of integration (%d)", method);
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_fwd_scan: bad sizes P=%d N=%d D=%d", P, N, D);
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_fwd_scan: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    VGPA_REQUIRE(A && b && m0 && S0 && Sigma && m_out && S_out, "vgpa_fwd_scan: null pointer argument");
    VGPA_REQUIRE(sigma_stride == 0 || sigma_stride == D * D, "vgpa_fwd_scan: sigma_stride must be 0 or D*D");
    if (P == 0) return VGPA_OK;
    if (D > 1 && !nd_fits(D, method, false))
        return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_fwd_scan: D=%d with method %d needs more than the 160 KB of LDS of a compute unit", D, method);
    g_lds_refused = false;
    hipStream_t st = vgpa::as_stream(stream);
    if (flags & VGPA_SCAN_PACKED) {
        if (!(method == VGPA_EULER && D == 40 && (flags & VGPA_SCAN_SYMMETRIC)))
            return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_fwd_scan: VGPA_SCAN_PACKED needs the Euler method, D == 40 and "
                              "VGPA_SCAN_SYMMETRIC (got method %d, D=%d, flags %d)", method, D, flags);
        const size_t lds = sizeof(double) * M40::LDS_DOUBLES;
        allow_lds(fwd_euler_sym_mfma40_kernel<true>, lds);
        if (g_lds_refused) return vgpa::fail(VGPA_ELAUNCH, "vgpa_fwd_scan: the LDS size this launch needs was refused");
        hipLaunchKernelGGL(fwd_euler_sym_mfma40_kernel<true>, dim3(P), dim3(M40::NT), lds, st, N, A, b, m0, S0, Sigma,
                           sigma_stride, dt, m_out, S_out);
        return vgpa::check_launch("vgpa_fwd_scan");
    }
    if (D == 1) {
        // one workgroup per problem, the steps as a parallel prefix computation
        const size_t lds = Prefix1D::fwd_lds_bytes();
#define LAUNCH1(M) hipLaunchKernelGGL(fwd_scan_1d_prefix_kernel<M>, dim3(P), dim3(Prefix1D::NT), lds, st, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out)
        switch (method) {
            case VGPA_EULER: LAUNCH1(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCH1(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCH1(VGPA_RK2); break;
            default: LAUNCH1(VGPA_RK4); break;
        }
#undef LAUNCH1
    } else {
        VGPA_DISPATCH_SCAN(launch_fwd, P, N, D, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out, st);
    }
    if (g_lds_refused) {
        (void)hipGetLastError();
        return vgpa::fail(VGPA_ELAUNCH, "vgpa_fwd_scan: the LDS size this launch needs was refused (D=%d)", D);
    }
    return vgpa::check_launch("vgpa_fwd_scan");
}

extern "C" int vgpa_bwd_scan(int method, int P, int N, int D, const double* A, const double* dEsde_dm,
                             const double* dEsde_dS, const double* dEobs_dm, const double* dEobs_dS,
                             const int* obs_flag, const double* jumpS, int jumpS_stride, double dt, int flags,
                             const double* lam_last, const double* Psi_last,
                             double* lam_out, double* Psi_out, void* stream) {
    VGPA_REQUIRE(method >= VGPA_EULER && method <= VGPA_RK4, "vgpa_bwd_scan: Unknown method of integration (%d)", method);
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_bwd_scan: bad sizes P=%d N=%d D=%d", P, N, D);
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_bwd_scan: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    VGPA_REQUIRE(A && dEsde_dm && dEsde_dS && dEobs_dm && lam_out && Psi_out, "vgpa_bwd_scan: null pointer argument");
    VGPA_REQUIRE(jumpS_stride == 0 || jumpS_stride == D * D, "vgpa_bwd_scan: jumpS_stride must be 0 or D*D");
    if (P == 0) return VGPA_OK;
    if (D > 1 && !nd_fits(D, method, false))
        return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_bwd_scan: D=%d with method %d needs more than the 160 KB of LDS of a compute unit", D, method);
    g_lds_refused = false;
    hipStream_t st = vgpa::as_stream(stream);
    if (D == 1) {
        const size_t lds = Prefix1D::bwd_lds_bytes();
#define LAUNCH1(M) do { allow_lds(bwd_scan_1d_prefix_kernel<M>, lds); hipLaunchKernelGGL(bwd_scan_1d_prefix_kernel<M>, dim3(P), dim3(Prefix1D::NT), lds, st, N, A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, obs_flag, jumpS, jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out); } while (0)
        switch (method) {
            case VGPA_EULER: LAUNCH1(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCH1(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCH1(VGPA_RK2); break;
            default: LAUNCH1(VGPA_RK4); break;
        }
#undef LAUNCH1
    } else {
        VGPA_DISPATCH_SCAN(launch_bwd, P, N, D, A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, obs_flag, jumpS,
                           jumpS_stride, dt, lam_last, Psi_last, lam_out, Psi_out, st);
    }
    if (g_lds_refused) {
        (void)hipGetLastError();
        return vgpa::fail(VGPA_ELAUNCH, "vgpa_bwd_scan: the LDS size this launch needs was refused (D=%d)", D);
    }
    return vgpa::check_launch("vgpa_bwd_scan");
}

extern "C" int vgpa_bwd_scan_grad_is_fused(int method, int P, int D, int scan_flags, int grad_flags, int dense_jumps,
                                           int explicit_edf, int edf_system) {
    // One workgroup per problem: below about 160 problems the chip is not full, a step's latency
    // and not the memory traffic sets the time, and the separate scan (shorter step) followed by the
    // fully parallel assembly kernel is faster (measured: 6.5 against 7.9 ms at 128 problems, 9.4
    // against 8.4 ms at 256, N = 2002).
    return P > 160 && (method == VGPA_EULER || method == VGPA_HEUN || method == VGPA_RK2) && D == 40 && (scan_flags & VGPA_SCAN_SYMMETRIC) &&
           (scan_flags & VGPA_SCAN_STATE_SYMMETRIC) && !dense_jumps && !explicit_edf && edf_system == VGPA_SYS_L96_UT &&
           (grad_flags & VGPA_GRAD_SIGINV_DIAG) && !(grad_flags & VGPA_GRAD_OUTER);
}

extern "C" int vgpa_bwd_scan_grad(int method, int P, int N, int D, const double* A, const double* b, const double* m,
                                  const double* S, const double* dEsde_dm, const double* dEsde_dS,
                                  const double* dEobs_dm, const double* dEobs_dS, const int* obs_flag,
                                  const double* jumpS, int jumpS_stride, const double* Ef, const double* Edf,
                                  int edf_system, const double* theta, int theta_stride, const double* SigInv,
                                  int sig_stride, double dt, int scan_flags, int grad_flags, const double* lam_last,
                                  const double* Psi_last, double* lam_out, double* Psi_out, double* gLA, double* gLb,
                                  void* stream) {
    VGPA_REQUIRE(method >= VGPA_EULER && method <= VGPA_RK4, "vgpa_bwd_scan_grad: Unknown method of integration (%d)", method);
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_bwd_scan_grad: bad sizes P=%d N=%d D=%d", P, N, D);
    VGPA_REQUIRE(A && b && m && S && dEsde_dm && dEsde_dS && dEobs_dm && Ef && SigInv && lam_out && Psi_out && gLA && gLb,
                 "vgpa_bwd_scan_grad: null pointer argument");
    const bool fused = vgpa_bwd_scan_grad_is_fused(method, P, D, scan_flags, grad_flags, dEobs_dS != nullptr,
                                                   Edf != nullptr, edf_system) != 0;
    const bool packed = (scan_flags & VGPA_SCAN_PACKED) != 0, packed_g = (scan_flags & VGPA_SCAN_PACKED_DEDS) != 0;
    if ((packed || packed_g) && !(fused && method == VGPA_EULER))
        return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_bwd_scan_grad: VGPA_SCAN_PACKED / VGPA_SCAN_PACKED_DEDS are accepted only "
                          "with the Euler method where vgpa_bwd_scan_grad_is_fused() holds");
    if (!fused && (scan_flags & VGPA_SCAN_PSI_ROW0))
        return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_bwd_scan_grad: VGPA_SCAN_PSI_ROW0 is accepted only where "
                          "vgpa_bwd_scan_grad_is_fused() holds (the assembly kernel needs every row of Psi)");
    if (!fused) {
        // every other configuration: the scan, then the assembly kernel
        const int rc = vgpa_bwd_scan(method, P, N, D, A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, obs_flag, jumpS,
                                     jumpS_stride, dt, scan_flags & VGPA_SCAN_SYMMETRIC, lam_last, Psi_last, lam_out,
                                     Psi_out, stream);
        if (rc != VGPA_OK) return rc;
        return vgpa_grad_assemble(P, N, D, A, b, m, S, lam_out, Psi_out, Ef, Edf, edf_system, theta, theta_stride, SigInv,
                                  sig_stride, dt, grad_flags, gLA, gLb, stream);
    }
    VGPA_REQUIRE(jumpS_stride == 0 || jumpS_stride == D * D, "vgpa_bwd_scan_grad: jumpS_stride must be 0 or D*D");
    VGPA_REQUIRE(sig_stride == 0 || sig_stride == D * D, "vgpa_bwd_scan_grad: sig_stride must be 0 or D*D");
    if (P == 0) return VGPA_OK;
    g_lds_refused = false;
    const int psi_all = (scan_flags & VGPA_SCAN_PSI_ROW0) ? 0 : 1;
    if (method != VGPA_EULER) {
        const size_t lds = sizeof(double) * M40FH::LDS_DOUBLES;
        auto launch = [&](auto kernel) {
            allow_lds(kernel, lds);
            if (g_lds_refused) return;
            hipLaunchKernelGGL(kernel, dim3(P), dim3(M40FH::NT), lds, vgpa::as_stream(stream), N, A, b, m, S, dEsde_dm,
                               dEsde_dS, dEobs_dm, obs_flag, jumpS, jumpS_stride, Ef, SigInv, sig_stride, dt, lam_last,
                               Psi_last, lam_out, Psi_out, psi_all, gLA, gLb);
        };
        if (method == VGPA_HEUN) launch(bwd_grad_stages2_sym_mfma40_kernel<VGPA_HEUN>);
        else launch(bwd_grad_stages2_sym_mfma40_kernel<VGPA_RK2>);
        if (g_lds_refused) return vgpa::fail(VGPA_ELAUNCH, "vgpa_bwd_scan_grad: the LDS size this launch needs was refused");
        return vgpa::check_launch("vgpa_bwd_scan_grad");
    }
    const size_t lds = sizeof(double) * M40F::LDS_DOUBLES;
    auto launch = [&](auto kernel) {
        allow_lds(kernel, lds);
        if (g_lds_refused) return;
        hipLaunchKernelGGL(kernel, dim3(P), dim3(M40F::NT), lds, vgpa::as_stream(stream), N, A, b, m, S, dEsde_dm, dEsde_dS,
                           dEobs_dm, obs_flag, jumpS, jumpS_stride, Ef, SigInv, sig_stride, dt, lam_last, Psi_last, lam_out,
                           Psi_out, psi_all, gLA, gLb);
    };
    if (packed) launch(bwd_grad_euler_sym_mfma40_kernel<true, true, true>);
    else if (packed_g) launch(bwd_grad_euler_sym_mfma40_kernel<false, true, false>);
    else launch(bwd_grad_euler_sym_mfma40_kernel<false, false, false>);
    if (g_lds_refused) return vgpa::fail(VGPA_ELAUNCH, "vgpa_bwd_scan_grad: the LDS size this launch needs was refused");
    return vgpa::check_launch("vgpa_bwd_scan_grad");
}

namespace {

// one thread per 16-byte piece of a packed matrix
template <bool TO_PACKED>
__global__ void __launch_bounds__(256) sym_rows_convert_kernel(size_t count, const double* __restrict__ src,
                                                               double* __restrict__ dst) {
    using L = vgpa::SymRows40;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count * L::PIECES) return;
    const size_t mtx = i / L::PIECES;
    const int e = 2 * (int)(i % L::PIECES);
    int r, c;
    L::coords(e, r, c);
    if (c >= L::D) return;                                   // padding
    const double* dense_in = src + mtx * (L::D * L::D);
    double* dense_out = dst + mtx * (L::D * L::D);
    if (TO_PACKED) {
        *reinterpret_cast<double2*>(dst + mtx * L::SIZE + e) = *reinterpret_cast<const double2*>(dense_in + r * L::D + c);
    } else {
        const double2 v = *reinterpret_cast<const double2*>(src + mtx * L::SIZE + e);
        *reinterpret_cast<double2*>(dense_out + r * L::D + c) = v;
        if (c < 16 * (r / 16)) {                             // left of the diagonal block: also the mirror image
            dense_out[c * L::D + r] = v.x;
            dense_out[(c + 1) * L::D + r] = v.y;
        }
    }
}

}  // namespace

extern "C" int vgpa_sym_pack(long long count, int D, const double* dense, double* packed, void* stream) {
    VGPA_REQUIRE(count >= 0 && dense && packed, "vgpa_sym_pack: bad arguments");
    if (D != 40) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_sym_pack: the packed layout exists for D == 40 only (got %d)", D);
    if (count == 0) return VGPA_OK;
    const size_t threads = (size_t)count * vgpa::SymRows40::PIECES;
    VGPA_REQUIRE((threads + 255) / 256 < 0x7fffffffull, "vgpa_sym_pack: too many matrices for one call");
    hipLaunchKernelGGL(sym_rows_convert_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       vgpa::as_stream(stream), (size_t)count, dense, packed);
    return vgpa::check_launch("vgpa_sym_pack");
}

extern "C" int vgpa_sym_unpack(long long count, int D, const double* packed, double* dense, void* stream) {
    VGPA_REQUIRE(count >= 0 && dense && packed, "vgpa_sym_unpack: bad arguments");
    if (D != 40) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_sym_unpack: the packed layout exists for D == 40 only (got %d)", D);
    if (count == 0) return VGPA_OK;
    const size_t threads = (size_t)count * vgpa::SymRows40::PIECES;
    VGPA_REQUIRE((threads + 255) / 256 < 0x7fffffffull, "vgpa_sym_unpack: too many matrices for one call");
    hipLaunchKernelGGL(sym_rows_convert_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       vgpa::as_stream(stream), (size_t)count, packed, dense);
    return vgpa::check_launch("vgpa_sym_unpack");
}

extern "C" int vgpa_sym_packed_size(int D) { return D == 40 ? vgpa::SymRows40::SIZE : -1; }

#ifdef VGPA_STAMPS
extern "C" int vgpa_debug_m40_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_m40_stamps), 128 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
