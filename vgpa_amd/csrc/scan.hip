// Forward moment scans m(t), S(t) and backward multiplier scans lam(t), Psi(t)
// for gfx950.
//
// The recurrences are sequential in time, so parallelism comes from the batch of
// independent problems (one workgroup per problem for D > 1, one lane per problem
// for D == 1) and from the D x D matrix products inside a step.
//
// n-D layout inside a workgroup: every thread owns one TI x TJ tile of the D x D
// state (S or Psi) and keeps it, and its Runge-Kutta increments, in registers for
// the whole scan.  Only the stage input X (what the other threads need for the
// matrix product) goes through LDS, in two alternating buffers so that one
// barrier per stage is enough.  A(t) is staged through LDS, prefetched one step
// ahead into registers; rows are padded by one double so that both row- and
// column-indexed reads are free of bank conflicts.
#include "common.h"

namespace {

constexpr int TI = 2;    // tile rows per thread
constexpr int TJ = 4;    // tile cols per thread
constexpr int PRE = 8;   // max A elements a thread stages per step (D*D / nthreads <= 8)

struct NdGeom {
    int D, LD, MS;        // dimension, padded leading dimension, padded matrix size
    int nthreads;
    int i0, j0;           // tile origin
    bool has_tile;
    int vi;               // row of the vector state owned by this thread, or -1
};

__device__ inline NdGeom make_geom(int D) {
    NdGeom g;
    g.D = D;
    g.LD = D + 1;
    g.MS = D * g.LD;
    g.nthreads = blockDim.x;
    const int ntj = (D + TJ - 1) / TJ;
    const int nti = (D + TI - 1) / TI;
    const int tile = threadIdx.x;
    g.has_tile = tile < nti * ntj;
    g.i0 = (tile / ntj) * TI;
    g.j0 = (tile % ntj) * TJ;
    const int v = g.nthreads - 1 - (int)threadIdx.x;   // vector rows live on the last D threads
    g.vi = (v < D) ? v : -1;
    return g;
}

__device__ inline int clampi(int x, int hi) { return x < hi ? x : hi; }

// Copy a dense D x D matrix from global memory into a padded LDS image.
__device__ inline void load_matrix_lds(const NdGeom& g, const double* __restrict__ src, double* dst) {
    const int DD = g.D * g.D;
    for (int e = threadIdx.x; e < DD; e += g.nthreads) dst[(e / g.D) * g.LD + (e % g.D)] = src[e];
}

__device__ inline void prefetch_matrix(const NdGeom& g, const double* __restrict__ src, double pre[PRE]) {
    const int DD = g.D * g.D;
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int e = threadIdx.x + q * g.nthreads;
        pre[q] = (e < DD) ? src[e] : 0.0;
    }
}

// Write the prefetched matrix into `dst` and, when `mid` is non-null, the midpoint
// 0.5*(other + new) into `mid` (each element is handled by the thread that staged it).
__device__ inline void commit_matrix(const NdGeom& g, const double pre[PRE], double* dst,
                                     const double* other, double* mid) {
    const int DD = g.D * g.D;
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int e = threadIdx.x + q * g.nthreads;
        if (e < DD) {
            const int o = (e / g.D) * g.LD + (e % g.D);
            dst[o] = pre[q];
            if (mid) mid[o] = 0.5 * (other[o] + pre[q]);
        }
    }
}

__device__ inline void load_tile(const NdGeom& g, const double* __restrict__ src, double t[TI][TJ]) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            const int i = g.i0 + a, j = g.j0 + c;
            t[a][c] = (g.has_tile && i < g.D && j < g.D) ? src[i * g.D + j] : 0.0;
        }
}

__device__ inline void store_tile_global(const NdGeom& g, double* __restrict__ dst, const double t[TI][TJ]) {
    if (!g.has_tile) return;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            const int i = g.i0 + a, j = g.j0 + c;
            if (i < g.D && j < g.D) dst[i * g.D + j] = t[a][c];
        }
}

__device__ inline void store_tile_lds(const NdGeom& g, double* dst, const double t[TI][TJ]) {
    if (!g.has_tile) return;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            const int i = g.i0 + a, j = g.j0 + c;
            if (i < g.D && j < g.D) dst[i * g.LD + j] = t[a][c];
        }
}

// k = Sig - Am*X - X*Am'   (core/fwd_ode.py:45)
__device__ inline void fwd_rhs_tile(const NdGeom& g, const double* Am, const double* X,
                                    const double sig[TI][TJ], double k[TI][TJ]) {
    if (!g.has_tile) return;
    int ri[TI], cj[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) ri[a] = clampi(g.i0 + a, g.D - 1) * g.LD;
#pragma unroll
    for (int c = 0; c < TJ; ++c) cj[c] = clampi(g.j0 + c, g.D - 1);
    double s1[TI][TJ] = {}, s2[TI][TJ] = {};
    for (int q = 0; q < g.D; ++q) {
        double a_iq[TI], x_iq[TI], x_qj[TJ], a_jq[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) { a_iq[a] = Am[ri[a] + q]; x_iq[a] = X[ri[a] + q]; }
#pragma unroll
        for (int c = 0; c < TJ; ++c) { x_qj[c] = X[q * g.LD + cj[c]]; a_jq[c] = Am[cj[c] * g.LD + q]; }
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                s1[a][c] = fma(a_iq[a], x_qj[c], s1[a][c]);
                s2[a][c] = fma(x_iq[a], a_jq[c], s2[a][c]);
            }
    }
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) k[a][c] = -s1[a][c] - s2[a][c] + sig[a][c];
}

// k = -G + X*Am + Am'*X   (core/bwd_ode.py:43)
__device__ inline void bwd_rhs_tile(const NdGeom& g, const double* Am, const double* X,
                                    const double G[TI][TJ], double k[TI][TJ]) {
    if (!g.has_tile) return;
    int ri[TI], ci[TI], cj[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) { ci[a] = clampi(g.i0 + a, g.D - 1); ri[a] = ci[a] * g.LD; }
#pragma unroll
    for (int c = 0; c < TJ; ++c) cj[c] = clampi(g.j0 + c, g.D - 1);
    double s1[TI][TJ] = {}, s2[TI][TJ] = {};
    for (int q = 0; q < g.D; ++q) {
        double x_iq[TI], a_qi[TI], a_qj[TJ], x_qj[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) { x_iq[a] = X[ri[a] + q]; a_qi[a] = Am[q * g.LD + ci[a]]; }
#pragma unroll
        for (int c = 0; c < TJ; ++c) { a_qj[c] = Am[q * g.LD + cj[c]]; x_qj[c] = X[q * g.LD + cj[c]]; }
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                s1[a][c] = fma(x_iq[a], a_qj[c], s1[a][c]);
                s2[a][c] = fma(a_qi[a], x_qj[c], s2[a][c]);
            }
    }
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) k[a][c] = -G[a][c] + s1[a][c] + s2[a][c];
}

// (Am x)[vi]
__device__ inline double matvec_row(const NdGeom& g, const double* Am, const double* x) {
    double acc = 0.0;
    const double* row = Am + g.vi * g.LD;
    for (int q = 0; q < g.D; ++q) acc = fma(row[q], x[q], acc);
    return acc;
}

#define TILE_FOR(a, c) \
    _Pragma("unroll") for (int a = 0; a < TI; ++a) _Pragma("unroll") for (int c = 0; c < TJ; ++c)

// ---------------------------------------------------------------------------
// forward scan, D > 1: one workgroup per problem.
// core/fwd_ode.py: euler_nD :87, heun_nD :181, rk2_nD :311, rk4_nD :426.
// ---------------------------------------------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(512)
fwd_scan_nd_kernel(int N, int D, const double* __restrict__ A, const double* __restrict__ b,
                   const double* __restrict__ m0, const double* __restrict__ S0,
                   const double* __restrict__ Sigma, int sigma_stride, double dt,
                   double* __restrict__ m_out, double* __restrict__ S_out) {
    extern __shared__ double lds[];
    const NdGeom g = make_geom(D);
    const int p = blockIdx.x;
    const size_t DD = (size_t)D * D;
    double* Abuf[2] = {lds, lds + g.MS};
    double* Amid = lds + 2 * g.MS;
    double* Xb[2] = {lds + 3 * g.MS, lds + 4 * g.MS};
    double* vec = lds + 5 * g.MS;
    double* bbuf[2] = {vec, vec + D};
    double* bmid = vec + 2 * D;
    double* xb[2] = {vec + 3 * D, vec + 4 * D};

    const double* Ap = A + (size_t)p * N * DD;
    const double* bp = b + (size_t)p * N * D;
    double* mo = m_out + (size_t)p * N * D;
    double* So = S_out + (size_t)p * N * DD;
    const double h = 0.5 * dt;

    double sig[TI][TJ], s[TI][TJ];
    load_tile(g, Sigma + (size_t)p * sigma_stride, sig);
    load_tile(g, S0 + (size_t)p * DD, s);
    double mv = (g.vi >= 0) ? m0[(size_t)p * D + g.vi] : 0.0;

    load_matrix_lds(g, Ap, Abuf[0]);
    store_tile_lds(g, Xb[0], s);
    store_tile_global(g, So, s);
    if (g.vi >= 0) {
        bbuf[0][g.vi] = bp[g.vi];
        xb[0][g.vi] = mv;
        mo[g.vi] = mv;
    }
    double pre[PRE];
    if (N > 1) prefetch_matrix(g, Ap + DD, pre);
    int cur = 0;   // A buffer holding A[t]
    int xin = 0;   // X buffer holding S[t] / m[t]

    for (int t = 0; t + 1 < N; ++t) {
        __syncthreads();   // previous step's LDS writes visible, previous reads retired
        const int nxt = cur ^ 1;
        commit_matrix(g, pre, Abuf[nxt], Abuf[cur], (METHOD == VGPA_RK2 || METHOD == VGPA_RK4) ? Amid : nullptr);
        if (g.vi >= 0) {
            const double bn = bp[(size_t)(t + 1) * D + g.vi];
            bbuf[nxt][g.vi] = bn;
            if (METHOD == VGPA_RK2 || METHOD == VGPA_RK4) bmid[g.vi] = 0.5 * (bbuf[cur][g.vi] + bn);
        }
        if (t + 2 < N) prefetch_matrix(g, Ap + (size_t)(t + 2) * DD, pre);

        double k1[TI][TJ], snew[TI][TJ];
        double mnew = 0.0;
        fwd_rhs_tile(g, Abuf[cur], Xb[xin], sig, k1);
        double f1 = 0.0;
        if (g.vi >= 0) f1 = -matvec_row(g, Abuf[cur], xb[xin]) + bbuf[cur][g.vi];

        if (METHOD == VGPA_EULER) {
            TILE_FOR(a, c) snew[a][c] = s[a][c] + k1[a][c] * dt;
            mnew = mv + f1 * dt;
            store_tile_lds(g, Xb[xin ^ 1], snew);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = mnew;
            xin ^= 1;
        } else if (METHOD == VGPA_HEUN) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] + k1[a][c] * dt;
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = mv + f1 * dt;
            __syncthreads();
            fwd_rhs_tile(g, Abuf[nxt], Xb[xin ^ 1], sig, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] + h * (k1[a][c] + k2[a][c]);
            if (g.vi >= 0) {
                const double f2 = -matvec_row(g, Abuf[nxt], xb[xin ^ 1]) + bbuf[nxt][g.vi];
                mnew = mv + h * (f1 + f2);
            }
            store_tile_lds(g, Xb[xin], snew);
            if (g.vi >= 0) xb[xin][g.vi] = mnew;
        } else if (METHOD == VGPA_RK2) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] + h * k1[a][c];
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = mv + h * f1;
            __syncthreads();
            fwd_rhs_tile(g, Amid, Xb[xin ^ 1], sig, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] + k2[a][c] * dt;
            if (g.vi >= 0) {
                const double f2 = -matvec_row(g, Amid, xb[xin ^ 1]) + bmid[g.vi];
                mnew = mv + f2 * dt;
            }
            store_tile_lds(g, Xb[xin], snew);
            if (g.vi >= 0) xb[xin][g.vi] = mnew;
        } else {   // RK4
            double L1[TI][TJ], L2[TI][TJ], L3[TI][TJ], L4[TI][TJ], tmp[TI][TJ];
            double K1 = f1 * dt, K2 = 0.0, K3 = 0.0, K4 = 0.0;
            TILE_FOR(a, c) { L1[a][c] = k1[a][c] * dt; tmp[a][c] = s[a][c] + 0.5 * L1[a][c]; }
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = mv + 0.5 * K1;
            __syncthreads();
            fwd_rhs_tile(g, Amid, Xb[xin ^ 1], sig, L2);
            TILE_FOR(a, c) { L2[a][c] *= dt; tmp[a][c] = s[a][c] + 0.5 * L2[a][c]; }
            if (g.vi >= 0) K2 = (-matvec_row(g, Amid, xb[xin ^ 1]) + bmid[g.vi]) * dt;
            store_tile_lds(g, Xb[xin], tmp);
            if (g.vi >= 0) xb[xin][g.vi] = mv + 0.5 * K2;
            __syncthreads();
            fwd_rhs_tile(g, Amid, Xb[xin], sig, L3);
            TILE_FOR(a, c) { L3[a][c] *= dt; tmp[a][c] = s[a][c] + L3[a][c]; }
            if (g.vi >= 0) K3 = (-matvec_row(g, Amid, xb[xin]) + bmid[g.vi]) * dt;
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = mv + K3;
            __syncthreads();
            fwd_rhs_tile(g, Abuf[nxt], Xb[xin ^ 1], sig, L4);
            TILE_FOR(a, c) {
                L4[a][c] *= dt;
                snew[a][c] = s[a][c] + (L1[a][c] + 2.0 * (L2[a][c] + L3[a][c]) + L4[a][c]) / 6.0;
            }
            if (g.vi >= 0) {
                K4 = (-matvec_row(g, Abuf[nxt], xb[xin ^ 1]) + bbuf[nxt][g.vi]) * dt;
                mnew = mv + (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            }
            store_tile_lds(g, Xb[xin], snew);
            if (g.vi >= 0) xb[xin][g.vi] = mnew;
        }
        TILE_FOR(a, c) s[a][c] = snew[a][c];
        store_tile_global(g, So + (size_t)(t + 1) * DD, s);
        if (g.vi >= 0) {
            mv = mnew;
            mo[(size_t)(t + 1) * D + g.vi] = mv;
        }
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------
// backward scan, D > 1: one workgroup per problem.
// core/bwd_ode.py: euler_nD :86, heun_nD :177, rk2_nD :305, rk4_nD :427.
// ---------------------------------------------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(512)
bwd_scan_nd_kernel(int N, int D, const double* __restrict__ A, const double* __restrict__ dEdm,
                   const double* __restrict__ dEdS, const double* __restrict__ jm,
                   const double* __restrict__ jS_dense, const int* __restrict__ obs_flag,
                   const double* __restrict__ jumpS, int jumpS_stride, double dt,
                   double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    extern __shared__ double lds[];
    const NdGeom g = make_geom(D);
    const int p = blockIdx.x;
    const size_t DD = (size_t)D * D;
    double* Abuf[2] = {lds, lds + g.MS};
    double* Amid = lds + 2 * g.MS;
    double* Xb[2] = {lds + 3 * g.MS, lds + 4 * g.MS};
    double* vec = lds + 5 * g.MS;
    double* xb[2] = {vec, vec + D};

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

    double s[TI][TJ];   // Psi[t]
    TILE_FOR(a, c) s[a][c] = 0.0;
    double lv = 0.0;    // lam[t][vi]
    store_tile_global(g, Po + (size_t)(N - 1) * DD, s);
    store_tile_lds(g, Xb[0], s);
    if (g.vi >= 0) {
        lo[(size_t)(N - 1) * D + g.vi] = 0.0;
        xb[0][g.vi] = 0.0;
    }
    load_matrix_lds(g, Ap + (size_t)(N - 1) * DD, Abuf[0]);
    double pre[PRE];
    if (N > 1) prefetch_matrix(g, Ap + (size_t)(N - 2) * DD, pre);
    double Gt[TI][TJ], Gp[TI][TJ];
    load_tile(g, gS + (size_t)(N - 1) * DD, Gt);
    double gmt = (g.vi >= 0) ? gm[(size_t)(N - 1) * D + g.vi] : 0.0;
    int cur = 0, xin = 0;

    for (int t = N - 1; t >= 1; --t) {
        __syncthreads();
        const int prv = cur ^ 1;   // buffer for A[t-1]
        commit_matrix(g, pre, Abuf[prv], Abuf[cur], (METHOD == VGPA_RK2 || METHOD == VGPA_RK4) ? Amid : nullptr);
        if (t >= 2) prefetch_matrix(g, Ap + (size_t)(t - 2) * DD, pre);
        load_tile(g, gS + (size_t)(t - 1) * DD, Gp);
        const double gmp = (g.vi >= 0) ? gm[(size_t)(t - 1) * D + g.vi] : 0.0;

        double k1[TI][TJ], snew[TI][TJ];
        double lnew = 0.0;
        bwd_rhs_tile(g, Abuf[cur], Xb[xin], Gt, k1);
        double f1 = 0.0;
        if (g.vi >= 0) f1 = -gmt + matvec_row(g, Abuf[cur], xb[xin]);

        if (METHOD == VGPA_EULER) {
            TILE_FOR(a, c) snew[a][c] = s[a][c] - k1[a][c] * dt;
            lnew = lv - f1 * dt;
        } else if (METHOD == VGPA_HEUN) {
            double tmp[TI][TJ], k2[TI][TJ];
            TILE_FOR(a, c) tmp[a][c] = s[a][c] - k1[a][c] * dt;
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = lv - f1 * dt;
            __syncthreads();
            bwd_rhs_tile(g, Abuf[prv], Xb[xin ^ 1], Gp, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] - h * (k1[a][c] + k2[a][c]);
            if (g.vi >= 0) {
                const double f2 = -gmp + matvec_row(g, Abuf[prv], xb[xin ^ 1]);
                lnew = lv - h * (f1 + f2);
            }
            xin ^= 1;   // final value goes to the buffer not read in the last stage
        } else if (METHOD == VGPA_RK2) {
            double tmp[TI][TJ], k2[TI][TJ], Gk[TI][TJ];
            TILE_FOR(a, c) { tmp[a][c] = s[a][c] - h * k1[a][c]; Gk[a][c] = 0.5 * (Gp[a][c] + Gt[a][c]); }
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = lv - h * f1;
            __syncthreads();
            bwd_rhs_tile(g, Amid, Xb[xin ^ 1], Gk, k2);
            TILE_FOR(a, c) snew[a][c] = s[a][c] - k2[a][c] * dt;
            if (g.vi >= 0) {
                const double f2 = -0.5 * (gmp + gmt) + matvec_row(g, Amid, xb[xin ^ 1]);
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
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = lv - 0.5 * K1;
            __syncthreads();
            bwd_rhs_tile(g, Amid, Xb[xin ^ 1], Gk, L2);
            TILE_FOR(a, c) { L2[a][c] *= dt; tmp[a][c] = s[a][c] - 0.5 * L2[a][c]; }
            if (g.vi >= 0) K2 = (-gmk + matvec_row(g, Amid, xb[xin ^ 1])) * dt;
            store_tile_lds(g, Xb[xin], tmp);
            if (g.vi >= 0) xb[xin][g.vi] = lv - 0.5 * K2;
            __syncthreads();
            bwd_rhs_tile(g, Amid, Xb[xin], Gk, L3);
            TILE_FOR(a, c) { L3[a][c] *= dt; tmp[a][c] = s[a][c] - L3[a][c]; }
            if (g.vi >= 0) K3 = (-gmk + matvec_row(g, Amid, xb[xin])) * dt;
            store_tile_lds(g, Xb[xin ^ 1], tmp);
            if (g.vi >= 0) xb[xin ^ 1][g.vi] = lv - K3;
            __syncthreads();
            bwd_rhs_tile(g, Abuf[prv], Xb[xin ^ 1], Gp, L4);
            TILE_FOR(a, c) {
                L4[a][c] *= dt;
                snew[a][c] = s[a][c] - (L1[a][c] + 2.0 * (L2[a][c] + L3[a][c]) + L4[a][c]) / 6.0;
            }
            if (g.vi >= 0) {
                K4 = (-gmp + matvec_row(g, Abuf[prv], xb[xin ^ 1])) * dt;
                lnew = lv - (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            }
            xin ^= 1;
        }
        // observation jumps at t-1 (core/bwd_ode.py:124,127)
        if (jSd) {
            double jd[TI][TJ];
            load_tile(g, jSd + (size_t)(t - 1) * DD, jd);
            TILE_FOR(a, c) snew[a][c] += jd[a][c];
        } else if (obs_flag && obs_flag[t - 1]) {
            TILE_FOR(a, c) snew[a][c] += jt[a][c];
        }
        if (g.vi >= 0) lnew += jmp[(size_t)(t - 1) * D + g.vi];

        // The last stage read Xb[xin_last]; the new state goes to the other buffer.
        // For EULER the last (only) stage read Xb[xin]; for the others xin was flipped
        // above to the buffer read last, so in every case write to xin ^ 1.
        xin ^= 1;
        TILE_FOR(a, c) { s[a][c] = snew[a][c]; Gt[a][c] = Gp[a][c]; }
        store_tile_lds(g, Xb[xin], s);
        store_tile_global(g, Po + (size_t)(t - 1) * DD, s);
        if (g.vi >= 0) {
            lv = lnew;
            gmt = gmp;
            xb[xin][g.vi] = lv;
            lo[(size_t)(t - 1) * D + g.vi] = lv;
        }
        cur = prv;
    }
}

// ---------------------------------------------------------------------------
// D == 1: one lane per problem.  core/fwd_ode.py: euler_1D :135, heun_1D :247,
// rk2_1D :367, rk4_1D :493;  core/bwd_ode.py: euler_1D :132, heun_1D :241,
// rk2_1D :367, rk4_1D :497.
// ---------------------------------------------------------------------------
template <int METHOD>
__global__ void fwd_scan_1d_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ b,
                                   const double* __restrict__ m0, const double* __restrict__ S0,
                                   const double* __restrict__ Sigma, int sigma_stride, double dt,
                                   double* __restrict__ m_out, double* __restrict__ S_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double* Ap = A + (size_t)p * N;
    const double* bp = b + (size_t)p * N;
    double* mo = m_out + (size_t)p * N;
    double* So = S_out + (size_t)p * N;
    const double Sig = Sigma[(size_t)p * sigma_stride];
    const double h = 0.5 * dt;
    double m = m0[p], S = S0[p];
    mo[0] = m;
    So[0] = S;
    double At = Ap[0], bt = bp[0];
    for (int t = 0; t + 1 < N; ++t) {
        const double An = Ap[t + 1], bn = bp[t + 1];
        if (METHOD == VGPA_EULER) {
            const double mn = m + (-At * m + bt) * dt;
            S = S + (-2 * At * S + Sig) * dt;
            m = mn;
        } else if (METHOD == VGPA_HEUN) {
            double ftp = -At * m + bt;
            double ftc = -An * (m + ftp * dt) + bn;
            const double mn = m + h * (ftp + ftc);
            ftp = -2 * At * S + Sig;
            ftc = -2 * An * (S + ftp * dt) + Sig;
            S = S + h * (ftp + ftc);
            m = mn;
        } else if (METHOD == VGPA_RK2) {
            const double Ak = 0.5 * (At + An), bk = 0.5 * (bt + bn);
            const double mn = m + (-Ak * (m + h * (-At * m + bt)) + bk) * dt;
            S = S + (-2 * Ak * (S + h * (-2 * At * S + Sig)) + Sig) * dt;
            m = mn;
        } else {
            const double Ak = 0.5 * (At + An), bk = 0.5 * (bt + bn);
            const double K1 = (-At * m + bt) * dt;
            const double K2 = (-Ak * (m + 0.5 * K1) + bk) * dt;
            const double K3 = (-Ak * (m + 0.5 * K2) + bk) * dt;
            const double K4 = (-An * (m + K3) + bn) * dt;
            const double L1 = (-2 * At * S + Sig) * dt;
            const double L2 = (-2 * Ak * (S + 0.5 * L1) + Sig) * dt;
            const double L3 = (-2 * Ak * (S + 0.5 * L2) + Sig) * dt;
            const double L4 = (-2 * An * (S + L3) + Sig) * dt;
            m = m + (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            S = S + (L1 + 2.0 * (L2 + L3) + L4) / 6.0;
        }
        mo[t + 1] = m;
        So[t + 1] = S;
        At = An;
        bt = bn;
    }
}

template <int METHOD>
__global__ void bwd_scan_1d_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                                   const double* __restrict__ dEdS, const double* __restrict__ jm,
                                   const double* __restrict__ jS_dense, const int* __restrict__ obs_flag,
                                   const double* __restrict__ jumpS, int jumpS_stride, double dt,
                                   double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double* Ap = A + (size_t)p * N;
    const double* gm = dEdm + (size_t)p * N;
    const double* gS = dEdS + (size_t)p * N;
    const double* jmp = jm + (size_t)p * N;
    const double* jSd = jS_dense ? jS_dense + (size_t)p * N : nullptr;
    const double jval = (!jSd && jumpS && obs_flag) ? jumpS[(size_t)p * jumpS_stride] : 0.0;
    double* lo = lam_out + (size_t)p * N;
    double* Po = Psi_out + (size_t)p * N;
    const double h = 0.5 * dt;
    double lam = 0.0, Psi = 0.0;
    lo[N - 1] = 0.0;
    Po[N - 1] = 0.0;
    double At = Ap[N - 1], gmt = gm[N - 1], gSt = gS[N - 1];
    for (int t = N - 1; t >= 1; --t) {
        const double Ak_ = Ap[t - 1], gmp = gm[t - 1], gSp = gS[t - 1];
        double ln, Pn;
        if (METHOD == VGPA_EULER) {
            ln = lam - (-gmt + At * lam) * dt;
            Pn = Psi - (-gSt + 2.0 * Psi * At) * dt;
        } else if (METHOD == VGPA_HEUN) {
            double ftp = -gmt + At * lam;
            double ftc = -gmp + Ak_ * (lam - ftp * dt);
            ln = lam - h * (ftp + ftc);
            ftp = -gSt + 2.0 * Psi * At;
            ftc = -gSp + 2.0 * (Psi - ftp * dt) * Ak_;
            Pn = Psi - h * (ftp + ftc);
        } else if (METHOD == VGPA_RK2) {
            const double Ak = 0.5 * (Ak_ + At), gmk = 0.5 * (gmp + gmt), gSk = 0.5 * (gSp + gSt);
            const double lamk = lam - h * (-gmt + At * lam);
            ln = lam - (-gmk + Ak * lamk) * dt;
            const double Psik = Psi - h * (-gSt + 2.0 * Psi * At);
            Pn = Psi - (-gSk + 2.0 * Psik * Ak) * dt;
        } else {
            const double Ak = 0.5 * (Ak_ + At), gmk = 0.5 * (gmp + gmt), gSk = 0.5 * (gSp + gSt);
            const double K1 = (-gmt + At * lam) * dt;
            const double K2 = (-gmk + Ak * (lam - 0.5 * K1)) * dt;
            const double K3 = (-gmk + Ak * (lam - 0.5 * K2)) * dt;
            const double K4 = (-gmp + Ak_ * (lam - K3)) * dt;
            ln = lam - (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
            const double L1 = (-gSt + 2.0 * Psi * At) * dt;
            const double L2 = (-gSk + 2.0 * (Psi - 0.5 * L1) * Ak) * dt;
            const double L3 = (-gSk + 2.0 * (Psi - 0.5 * L2) * Ak) * dt;
            const double L4 = (-gSp + 2.0 * (Psi - L3) * Ak_) * dt;
            Pn = Psi - (L1 + 2.0 * (L2 + L3) + L4) / 6.0;
        }
        ln += jmp[t - 1];
        if (jSd) Pn += jSd[t - 1];
        else if (obs_flag && obs_flag[t - 1]) Pn += jval;
        lam = ln;
        Psi = Pn;
        lo[t - 1] = lam;
        Po[t - 1] = Psi;
        At = Ak_;
        gmt = gmp;
        gSt = gSp;
    }
}

int nd_threads(int D) {
    const int ntiles = ((D + TI - 1) / TI) * ((D + TJ - 1) / TJ);
    int n = ntiles > D ? ntiles : D;
    n = vgpa::round_up(n, 64);
    return n < 64 ? 64 : n;
}

size_t nd_lds_bytes(int D) { return sizeof(double) * ((size_t)5 * D * (D + 1) + 5 * D); }

// Dynamic LDS above 64 KiB has to be requested per kernel.
template <typename K>
void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int vgpa_fwd_scan(int method, int P, int N, int D, const double* A, const double* b,
                             const double* m0, const double* S0, const double* Sigma, int sigma_stride,
                             double dt, double* m_out, double* S_out, void* stream) {
    VGPA_REQUIRE(method >= VGPA_EULER && method <= VGPA_RK4, "vgpa_fwd_scan: Unknown method of integration (%d)", method);
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_fwd_scan: bad sizes P=%d N=%d D=%d", P, N, D);
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_fwd_scan: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    VGPA_REQUIRE(A && b && m0 && S0 && Sigma && m_out && S_out, "vgpa_fwd_scan: null pointer argument");
    VGPA_REQUIRE(sigma_stride == 0 || sigma_stride == D * D, "vgpa_fwd_scan: sigma_stride must be 0 or D*D");
    if (P == 0) return VGPA_OK;
    hipStream_t st = vgpa::as_stream(stream);
    if (D == 1) {
        const int bs = 64, grid = (P + bs - 1) / bs;
#define LAUNCH1(M) hipLaunchKernelGGL(fwd_scan_1d_kernel<M>, dim3(grid), dim3(bs), 0, st, P, N, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out)
        switch (method) {
            case VGPA_EULER: LAUNCH1(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCH1(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCH1(VGPA_RK2); break;
            default: LAUNCH1(VGPA_RK4); break;
        }
#undef LAUNCH1
    } else {
        const int bs = nd_threads(D);
        const size_t lds = nd_lds_bytes(D);
#define LAUNCHN(M) allow_lds(fwd_scan_nd_kernel<M>, lds); hipLaunchKernelGGL(fwd_scan_nd_kernel<M>, dim3(P), dim3(bs), lds, st, N, D, A, b, m0, S0, Sigma, sigma_stride, dt, m_out, S_out)
        switch (method) {
            case VGPA_EULER: LAUNCHN(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCHN(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCHN(VGPA_RK2); break;
            default: LAUNCHN(VGPA_RK4); break;
        }
#undef LAUNCHN
    }
    return vgpa::check_launch("vgpa_fwd_scan");
}

extern "C" int vgpa_bwd_scan(int method, int P, int N, int D, const double* A, const double* dEsde_dm,
                             const double* dEsde_dS, const double* dEobs_dm, const double* dEobs_dS,
                             const int* obs_flag, const double* jumpS, int jumpS_stride, double dt,
                             double* lam_out, double* Psi_out, void* stream) {
    VGPA_REQUIRE(method >= VGPA_EULER && method <= VGPA_RK4, "vgpa_bwd_scan: Unknown method of integration (%d)", method);
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_bwd_scan: bad sizes P=%d N=%d D=%d", P, N, D);
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_bwd_scan: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    VGPA_REQUIRE(A && dEsde_dm && dEsde_dS && dEobs_dm && lam_out && Psi_out, "vgpa_bwd_scan: null pointer argument");
    VGPA_REQUIRE(jumpS_stride == 0 || jumpS_stride == D * D, "vgpa_bwd_scan: jumpS_stride must be 0 or D*D");
    if (P == 0) return VGPA_OK;
    hipStream_t st = vgpa::as_stream(stream);
    if (D == 1) {
        const int bs = 64, grid = (P + bs - 1) / bs;
#define LAUNCH1(M) hipLaunchKernelGGL(bwd_scan_1d_kernel<M>, dim3(grid), dim3(bs), 0, st, P, N, A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, obs_flag, jumpS, jumpS_stride, dt, lam_out, Psi_out)
        switch (method) {
            case VGPA_EULER: LAUNCH1(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCH1(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCH1(VGPA_RK2); break;
            default: LAUNCH1(VGPA_RK4); break;
        }
#undef LAUNCH1
    } else {
        const int bs = nd_threads(D);
        const size_t lds = nd_lds_bytes(D);
#define LAUNCHN(M) allow_lds(bwd_scan_nd_kernel<M>, lds); hipLaunchKernelGGL(bwd_scan_nd_kernel<M>, dim3(P), dim3(bs), lds, st, N, D, A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, obs_flag, jumpS, jumpS_stride, dt, lam_out, Psi_out)
        switch (method) {
            case VGPA_EULER: LAUNCHN(VGPA_EULER); break;
            case VGPA_HEUN: LAUNCHN(VGPA_HEUN); break;
            case VGPA_RK2: LAUNCHN(VGPA_RK2); break;
            default: LAUNCHN(VGPA_RK4); break;
        }
#undef LAUNCHN
    }
    return vgpa::check_launch("vgpa_bwd_scan");
}
