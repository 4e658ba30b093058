// Time reductions, observation terms, initial-state KL energy and the assembly of
// dL/dA, dL/db, gfx950.
#include "common.h"
#include "sym_rows.h"
#include <mutex>
#include "linalg.h"

#include <cmath>

namespace {

// ---------------------------------------------------------------------------
// out[p,c] = scale * sum_t w[t] x[p,t,c]     (auxiliary/numerical.py:231-284 as a
// weighted sum).  One workgroup per problem; lanes run over the contiguous (t,c)
// plane, partial sums are combined in a fixed order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
reduce_energy_kernel(int N, int C, const double* __restrict__ x, const double* __restrict__ w,
                     double scale, double* __restrict__ out) {
    __shared__ double red[256];
    const int p = blockIdx.x;
    const int T = 256 / C;            // time lanes per column (C <= 256)
    const int tid = threadIdx.x;
    const int c = tid % C, t0 = tid / C;
    double acc = 0.0;
    if (tid < T * C) {
        const double* xp = x + (size_t)p * N * C;
        for (int t = t0; t < N; t += T) acc = fma(w[t], xp[(size_t)t * C + c], acc);
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < C) {
        double s = 0.0;
        for (int k = 0; k < T; ++k) s += red[k * C + tid];
        out[(size_t)p * C + tid] = scale * s;
    }
}

// ---------------------------------------------------------------------------
// Observation terms, D > 1: one wavefront per (observation, problem); the energy
// contributions go to work[p][n] and are summed in a fixed order by
// obs_finish_kernel.  core/likelihood.py@108-179.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
obs_terms_nd_kernel(int N, int D, int M, const double* __restrict__ m, const double* __restrict__ S,
                    const int* __restrict__ obs_idx, const double* __restrict__ obsY, int obsY_stride,
                    const double* __restrict__ H, const double* __restrict__ Ri, const double* __restrict__ Qi,
                    double* __restrict__ work, double* __restrict__ dEobs_dm) {
    extern __shared__ double lds[];
    double* Wv = lds;          // W_n = y_n - m(t_n) H
    double* Tv = lds + D;      // Ri W_n
    const int n = blockIdx.x, p = blockIdx.y, l = threadIdx.x;
    const size_t DD = (size_t)D * D;
    const double* Y = obsY + (size_t)p * obsY_stride;
    const int tn = obs_idx[n];
    const double* mt = m + ((size_t)p * N + tn) * D;
    if (l < D) {
        double hm = 0.0;
        for (int i = 0; i < D; ++i) hm = fma(mt[i], H[i * D + l], hm);           // (m H)[l]
        Wv[l] = Y[(size_t)n * D + l] - hm;
    }
    __syncthreads();
    double e_l = 0.0;
    if (l < D) {
        double z = 0.0, t = 0.0;
        for (int j = 0; j < D; ++j) {
            z = fma(Wv[j], Qi[l * D + j], z);                                     // (W Qi')[l]
            t = fma(Ri[l * D + j], Wv[j], t);                                     // (Ri W)[l]
        }
        Tv[l] = t;
        e_l = z * z + Ri[l * D + l] * S[((size_t)p * N + tn) * DD + (size_t)l * D + l];
    }
    __syncthreads();
    if (l < D) {
        double gsum = 0.0;
        for (int k = 0; k < D; ++k) gsum = fma(H[k * D + l], Tv[k], gsum);       // (H' Ri W)[l]
        dEobs_dm[((size_t)p * N + tn) * D + l] = -gsum;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e_l += __shfl_xor(e_l, off, 64);
    if (l == 0) work[(size_t)p * M + n] = e_l;
}

// The same with the three D x D operators in LDS (rows padded by one double, so that lane l reading row l
// is free of bank conflicts): four wavefronts per workgroup, OBS_PER_WG observations of one problem per
// workgroup.  In the kernel above every lane walks a row of Ri and of Qi in global memory (forty cache lines
// per load instruction at D = 40) and every observation is a workgroup of its own.  Same products and sums
// in the same order: the same bits.
constexpr int OBS_PER_WG = 8;
__global__ void __launch_bounds__(256)
obs_terms_nd_lds_kernel(int N, int D, int M, const double* __restrict__ m, const double* __restrict__ S,
                        const int* __restrict__ obs_idx, const double* __restrict__ obsY, int obsY_stride,
                        const double* __restrict__ H, const double* __restrict__ Ri, const double* __restrict__ Qi,
                        double* __restrict__ work, double* __restrict__ dEobs_dm, int packed) {
    // packed (D = 40): S in the packed symmetric layout of sym_rows.h; only its diagonal is read here
    extern __shared__ double lds[];
    const int LD = D + 1, MS = D * LD;
    double* Hs = lds;
    double* Rs = Hs + MS;
    double* Qs = Rs + MS;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double* Wv = Qs + MS + w * 2 * D;          // W_n = y_n - m(t_n) H
    double* Tv = Wv + D;                       // Ri W_n
    for (int q = threadIdx.x; q < D * D; q += 256) {
        const int r = q / D, c = q - r * D;
        Hs[r * LD + c] = H[q];
        Rs[r * LD + c] = Ri[q];
        Qs[r * LD + c] = Qi[q];
    }
    __syncthreads();
    const int p = blockIdx.y;
    const size_t DD = packed ? (size_t)vgpa::SymRows40::SIZE : (size_t)D * D;
    const size_t sdiag = packed ? (size_t)(vgpa::SymRows40::row_offset(min(l, D - 1)) + l) : (size_t)l * D + l;
    const double* Y = obsY + (size_t)p * obsY_stride;
    for (int n = blockIdx.x * OBS_PER_WG + w; n < M && n < (blockIdx.x + 1) * OBS_PER_WG; n += 4) {
        const int tn = obs_idx[n];
        const double* mt = m + ((size_t)p * N + tn) * D;
        if (l < D) {
            double hm = 0.0;
            for (int i = 0; i < D; ++i) hm = fma(mt[i], Hs[i * LD + l], hm);
            Wv[l] = Y[(size_t)n * D + l] - hm;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double e_l = 0.0;
        if (l < D) {
            double z = 0.0, t = 0.0;
            for (int j = 0; j < D; ++j) {
                z = fma(Wv[j], Qs[l * LD + j], z);
                t = fma(Rs[l * LD + j], Wv[j], t);
            }
            Tv[l] = t;
            e_l = z * z + Rs[l * LD + l] * S[((size_t)p * N + tn) * DD + sdiag];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (l < D) {
            double gsum = 0.0;
            for (int k = 0; k < D; ++k) gsum = fma(Hs[k * LD + l], Tv[k], gsum);
            dEobs_dm[((size_t)p * N + tn) * D + l] = -gsum;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e_l += __shfl_xor(e_l, off, 64);
        if (l == 0) work[(size_t)p * M + n] = e_l;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // Wv, Tv are rewritten by the next observation
    }
}

// The same for D = 2, 3 with one LANE per (observation, problem): a wavefront per 3-vector leaves 61 lanes
// idle and a grid of thousands of problems costs hundreds of thousands of one-wavefront workgroups.  Same
// products and sums in the same order (the sum over components in the order of the shuffle tree above:
// (e_0 + e_2) + e_1), so the results are the same bits.
template <int D>
__global__ void __launch_bounds__(256)
obs_terms_small_kernel(int P, int N, int M, const double* __restrict__ m, const double* __restrict__ S,
                       const int* __restrict__ obs_idx, const double* __restrict__ obsY, int obsY_stride,
                       const double* __restrict__ H, const double* __restrict__ Ri, const double* __restrict__ Qi,
                       double* __restrict__ work, double* __restrict__ dEobs_dm) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)P * M) return;
    const int p = (int)(q / M), n = (int)(q % M);
    const double* Y = obsY + (size_t)p * obsY_stride + (size_t)n * D;
    const int tn = obs_idx[n];
    const double* mt = m + ((size_t)p * N + tn) * D;
    const double* St = S + ((size_t)p * N + tn) * D * D;
    double Wv[D], Tv[D], e[D];
#pragma unroll
    for (int l = 0; l < D; ++l) {
        double hm = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) hm = fma(mt[i], H[i * D + l], hm);
        Wv[l] = Y[l] - hm;
    }
#pragma unroll
    for (int l = 0; l < D; ++l) {
        double z = 0.0, t = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            z = fma(Wv[j], Qi[l * D + j], z);
            t = fma(Ri[l * D + j], Wv[j], t);
        }
        Tv[l] = t;
        e[l] = z * z + Ri[l * D + l] * St[l * D + l];
    }
#pragma unroll
    for (int l = 0; l < D; ++l) {
        double gsum = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) gsum = fma(H[k * D + l], Tv[k], gsum);
        dEobs_dm[((size_t)p * N + tn) * D + l] = -gsum;
    }
    work[q] = (D == 3) ? (e[0] + e[D - 1]) + e[1] : e[0] + e[1];
}

// jumpS = 0.5 H' Ri H (one workgroup; the D x D product Ri H goes through LDS).
__global__ void __launch_bounds__(256)
obs_jump_kernel(int D, const double* __restrict__ H, const double* __restrict__ Ri, double* __restrict__ jumpS) {
    extern __shared__ double lds[];   // RH[D][D]
    const int DD = D * D;
    for (int q = threadIdx.x; q < DD; q += blockDim.x) {
        const int a = q / D, j = q % D;
        double rh = 0.0;
        for (int c = 0; c < D; ++c) rh = fma(Ri[a * D + c], H[c * D + j], rh);
        lds[q] = rh;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < DD; q += blockDim.x) {
        const int i = q / D, j = q % D;
        double s = 0.0;
        for (int a = 0; a < D; ++a) s = fma(H[a * D + i], lds[a * D + j], s);
        jumpS[q] = 0.5 * s;
    }
}

// Eobs[p] = 0.5 (sum_n work[p][n] + logconst), observations in index order.
__global__ void obs_finish_kernel(int P, int M, const double* __restrict__ work, double logconst,
                                  double* __restrict__ Eobs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double s = 0.0;
    for (int n = 0; n < M; ++n) s += work[(size_t)p * M + n];
    Eobs[p] = 0.5 * (s + logconst);
}

// D == 1: one lane per problem.  core/likelihood.py@52-105 (the energy uses
// Y^2 - 2 Y m + <x^2> without H, the jump uses W = Y - H m; both as written).
__global__ void obs_terms_1d_kernel(int P, int N, int M, const double* __restrict__ m, const double* __restrict__ S,
                                    const int* __restrict__ obs_idx, const double* __restrict__ obsY, int obsY_stride,
                                    const double* __restrict__ Hp, const double* __restrict__ Rip, double logconst,
                                    double* __restrict__ Eobs, double* __restrict__ dEobs_dm,
                                    double* __restrict__ jumpS) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double H = Hp[0], Ri = Rip[0];
    const double* Y = obsY + (size_t)p * obsY_stride;
    double s = 0.0;
    for (int n = 0; n < M; ++n) {
        const int tn = obs_idx[n];
        const double mt = m[(size_t)p * N + tn], St = S[(size_t)p * N + tn], y = Y[n];
        s += y * y - 2 * y * mt + (mt * mt + St);
        dEobs_dm[(size_t)p * N + tn] = -Ri * (y - H * mt);
    }
    Eobs[p] = 0.5 * Ri * s + 0.5 * logconst;
    if (p == 0 && jumpS) jumpS[0] = 0.5 * Ri;
}


// ---------------------------------------------------------------------------
// KL0, D > 1: one wavefront per problem, matrices in LDS.  core/kl0.py:72-99.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
kl0_nd_kernel(int D, const double* __restrict__ m0, const double* __restrict__ S0,
              const double* __restrict__ mu0, int mu0_stride, const double* __restrict__ tau0, int tau0_stride,
              int m0_is_row, double* __restrict__ KL0) {
    extern __shared__ double lds[];
    const int LD = D + 1, MS = D * LD;
    double* Ma = lds;             // work copy
    double* S0i = lds + MS;       // inv(S0)
    double* T0i = lds + 2 * MS;   // inv(tau0)
    double* Pm = lds + 3 * MS;    // tau0 * inv(S0), then its Cholesky factor
    double* z = lds + 4 * MS;
    const int p = blockIdx.x, l = threadIdx.x;
    const size_t DD = (size_t)D * D;
    const double* S0p = S0 + (size_t)p * DD;
    const double* t0p = tau0 + (size_t)p * tau0_stride;
    for (int q = l; q < (int)DD; q += 64) Ma[(q / D) * LD + (q % D)] = S0p[q];
    if (l < D) z[l] = m0[(size_t)p * D + l] - mu0[(size_t)p * mu0_stride + l];
    __syncthreads();
    vgpa::gj_inverse(Ma, S0i, D, LD, l);
    for (int q = l; q < (int)DD; q += 64) Ma[(q / D) * LD + (q % D)] = t0p[q];
    __syncthreads();
    vgpa::gj_inverse(Ma, T0i, D, LD, l);
    // Pm = tau0 * S0i
    for (int q = l; q < (int)DD; q += 64) {
        const int i = q / D, j = q % D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = fma(t0p[i * D + k], S0i[k * LD + j], s);
        Pm[i * LD + j] = s;
    }
    __syncthreads();
    // log det via Cholesky of the lower triangle (auxiliary/numerical.py:86)
    double logdet = 0.0;
    bool ok = true;
    for (int j = 0; j < D; ++j) {
        const double djj = Pm[j * LD + j];
        if (!(djj > 0.0)) { ok = false; break; }
        const double d = sqrt(djj);
        logdet += log(d);
        double lij = 0.0;
        if (l < D && l > j) { lij = Pm[l * LD + j] / d; }
        __syncthreads();
        if (l < D && l > j) Pm[l * LD + j] = lij;
        __syncthreads();
        for (int k = j + 1; k < D; ++k) {
            const double lkj = Pm[k * LD + j];
            if (l < D && l >= k) Pm[l * LD + k] = fma(-lij, lkj, Pm[l * LD + k]);
        }
        __syncthreads();
    }
    // trace(tau0i (Z + S0 - tau0)),  Z = z z' (row mean) or |z|^2 in every entry
    double zz = 0.0;
    for (int k = 0; k < D; ++k) zz = fma(z[k], z[k], zz);
    double tr = 0.0;
    for (int q = l; q < (int)DD; q += 64) {
        const int i = q / D, j = q % D;
        const double Z = m0_is_row ? z[j] * z[i] : zz;
        tr = fma(T0i[i * LD + j], Z + S0p[j * D + i] - t0p[j * D + i], tr);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tr += __shfl_xor(tr, off, 64);
    if (l == 0) KL0[p] = ok ? 0.5 * (2.0 * logdet + tr) : __builtin_nan("");
}

// core/kl0.py:49-69 (note -log(S0), not -0.5 log(S0)).
__global__ void kl0_1d_kernel(int P, const double* __restrict__ m0, const double* __restrict__ S0,
                              const double* __restrict__ mu0, int mu0_stride, const double* __restrict__ tau0,
                              int tau0_stride, double* __restrict__ KL0) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double t0 = tau0[(size_t)p * tau0_stride];
    const double z0 = m0[p] - mu0[(size_t)p * mu0_stride];
    KL0[p] = -log(S0[p]) - 0.5 * (1.0 - log(t0)) + 0.5 / t0 * (z0 * z0 + S0[p]);
}

// <df/dx>[i][j] derived from the mean instead of read from memory.
// Lorenz-96: row i is filled in the order [i] = -1, [i+1] = m[i-1], [i-2] = -m[i-1],
// [i-1] = m[i+1] - m[i-2] (dynamics/sys_lorenz_96_ut.py:317-328; later entries win
// when indices collide for D < 4).  Lorenz-63: dynamics/sys_lorenz_63.py:227.
__device__ inline double edf_implicit(int system, int D, int i, int j, const double* mv, const double* th) {
    if (system == VGPA_SYS_L96_UT) {
        const int jf1 = (i + 1) % D, jb1 = (i + D - 1) % D, jb2 = (i + 2 * D - 2) % D;
        double v = 0.0;
        if (j == i) v = -1.0;
        if (j == jf1) v = mv[jb1];
        if (j == jb2) v = -mv[jb1];
        if (j == jb1) v = mv[jf1] - mv[jb2];
        return v;
    }
    // Lorenz-63 (analytic or unscented): th = (sigma, rho, beta)
    switch (i * 3 + j) {
        case 0: return -th[0];
        case 1: return th[0];
        case 2: return 0.0;
        case 3: return th[1] - mv[2];
        case 4: return -1.0;
        case 5: return -mv[0];
        case 6: return mv[1];
        case 7: return mv[0];
        default: return -th[2];
    }
}

// ---------------------------------------------------------------------------
// Gradient assembly, D > 1: one workgroup per (problem, step).
// core/gradL_Ab.py:156-177.  gLA = dt ( [SigInv (Edf + A) - 2 Psi] S - s1 - s2 ).
// ---------------------------------------------------------------------------
constexpr int GI = 2, GJ = 4;

__global__ void __launch_bounds__(512)
grad_assemble_nd_kernel(int N, int D, const double* __restrict__ A, const double* __restrict__ b,
                        const double* __restrict__ m, const double* __restrict__ S,
                        const double* __restrict__ lam, const double* __restrict__ Psi,
                        const double* __restrict__ Ef, const double* __restrict__ Edf, int edf_system,
                        const double* __restrict__ theta, int theta_stride,
                        const double* __restrict__ SigInv, int sig_stride, int sig_diag, double dt, int flags,
                        double* __restrict__ gLA, double* __restrict__ gLb) {
    extern __shared__ double lds[];
    const int LD = D + 1, MS = D * LD;
    double* Xm = lds;             // A, then SigInv (Edf + A) - 2 Psi
    double* Sm = lds + MS;        // S(t)
    double* Ym = lds + 2 * MS;    // Edf + A (general SigInv only)
    double* mv = lds + 3 * MS;
    double* uv = mv + D;          // -Ef - A m + b
    double* dbv = uv + D;
    double* lv = dbv + D;
    double* sc = lv + D;          // two scalars
    const size_t e = blockIdx.x;  // p * N + t
    const int p = (int)(e / N);
    const int tid = threadIdx.x, nth = blockDim.x;
    const size_t DD = (size_t)D * D;
    const double* Ae = A + e * DD;
    const double* Se = S + e * DD;
    const double* Pe = Psi + e * DD;
    const double* Ee = Edf ? Edf + e * DD : nullptr;
    const double* th = theta ? theta + (size_t)p * theta_stride : nullptr;
    const double* SI = SigInv + (size_t)p * sig_stride;
    for (int q = tid; q < (int)DD; q += nth) {
        const int o = (q / D) * LD + (q % D);
        Xm[o] = Ae[q];
        Sm[o] = Se[q];
    }
    if (tid < D) {
        mv[tid] = m[e * D + tid];
        lv[tid] = lam[e * D + tid];
    }
    __syncthreads();
    if (tid < D) {
        double am = 0.0;
        for (int k = 0; k < D; ++k) am = fma(Xm[tid * LD + k], mv[k], am);
        uv[tid] = -Ef[e * D + tid] - am + b[e * D + tid];
    }
    __syncthreads();
    if (tid < D) {
        double d;
        if (sig_diag) {
            d = uv[tid] * SI[(size_t)tid * D + tid];
        } else {
            d = 0.0;
            for (int i = 0; i < D; ++i) d = fma(uv[i], SI[(size_t)tid * D + i], d);
        }
        dbv[tid] = d;
        gLb[e * D + tid] = dt * (d + lv[tid]);
    }
    __syncthreads();
    if (tid == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < D; ++i) {
            s1 = fma(dbv[i], mv[i], s1);
            s2 = fma(lv[i], mv[i], s2);
        }
        sc[0] = s1;
        sc[1] = s2;
    }
    // X = SigInv (Edf + A) - 2 Psi
    if (sig_diag) {
        for (int q = tid; q < (int)DD; q += nth) {
            const int i = q / D, o = i * LD + (q % D);
            const double ed = Ee ? Ee[q] : edf_implicit(edf_system, D, i, q % D, mv, th);
            Xm[o] = SI[(size_t)i * D + i] * (ed + Xm[o]) - 2.0 * Pe[q];
        }
    } else {
        for (int q = tid; q < (int)DD; q += nth) {
            const int o = (q / D) * LD + (q % D);
            Ym[o] = (Ee ? Ee[q] : edf_implicit(edf_system, D, q / D, q % D, mv, th)) + Xm[o];
        }
        __syncthreads();
        for (int q = tid; q < (int)DD; q += nth) {
            const int i = q / D, j = q % D;
            double s = 0.0;
            for (int k = 0; k < D; ++k) s = fma(SI[(size_t)i * D + k], Ym[k * LD + j], s);
            Xm[i * LD + j] = s - 2.0 * Pe[q];
        }
    }
    __syncthreads();
    const int ntj = (D + GJ - 1) / GJ, nti = (D + GI - 1) / GI;
    if (tid < nti * ntj) {
        const int i0 = (tid / ntj) * GI, j0 = (tid % ntj) * GJ;
        int ri[GI], cj[GJ];
#pragma unroll
        for (int a = 0; a < GI; ++a) ri[a] = (i0 + a < D ? i0 + a : D - 1) * LD;
#pragma unroll
        for (int c = 0; c < GJ; ++c) cj[c] = (j0 + c < D ? j0 + c : D - 1);
        double acc[GI][GJ] = {};
        for (int k = 0; k < D; ++k) {
            double xv[GI], sv[GJ];
#pragma unroll
            for (int a = 0; a < GI; ++a) xv[a] = Xm[ri[a] + k];
#pragma unroll
            for (int c = 0; c < GJ; ++c) sv[c] = Sm[k * LD + cj[c]];
#pragma unroll
            for (int a = 0; a < GI; ++a)
#pragma unroll
                for (int c = 0; c < GJ; ++c) acc[a][c] = fma(xv[a], sv[c], acc[a][c]);
        }
        const double s1 = sc[0], s2 = sc[1];
#pragma unroll
        for (int a = 0; a < GI; ++a)
#pragma unroll
            for (int c = 0; c < GJ; ++c) {
                const int i = i0 + a, j = j0 + c;
                if (i < D && j < D) {
                    double corr;
                    if (flags & VGPA_GRAD_OUTER) corr = dbv[i] * mv[j] + lv[i] * mv[j];
                    else corr = s1 + s2;
                    gLA[e * DD + (size_t)i * D + j] = dt * (acc[a][c] - corr);
                }
            }
    }
}

#define VGPA_SMALL_D_GRAD
#include "small_d.h"

// Compile-time-D version for diagonal SigInv: 16-byte LDS reads, unrolled product,
// 16-byte global stores.  EDF_IMPLICIT derives <df/dx> of Lorenz-96 from the mean.
template <int D, bool EDF_IMPLICIT>
__global__ void __launch_bounds__(256)
grad_assemble_tiled_kernel(int N, const double* __restrict__ A, const double* __restrict__ b,
                           const double* __restrict__ m, const double* __restrict__ S,
                           const double* __restrict__ lam, const double* __restrict__ Psi,
                           const double* __restrict__ Ef, const double* __restrict__ Edf,
                           const double* __restrict__ SigInv, int sig_stride, double dt, int flags,
                           double* __restrict__ gLA, double* __restrict__ gLb) {
    constexpr int LD = D + 2, MS = D * LD, DD = D * D, NT = 256;
    constexpr int NTJ = D / GJ, NTI = D / GI;
    static_assert(NTI * NTJ <= NT && D % 4 == 0, "one 2x4 tile per thread");
    extern __shared__ double lds[];
    double* Xm = lds;             // A, then diag(SigInv) (Edf + A) - 2 Psi
    double* Sm = lds + MS;        // S(t)
    double* mv = lds + 2 * MS;
    double* lv = mv + D;
    double* dbv = lv + D;
    double* siv = dbv + D;
    double* sc = siv + D;         // s1 + s2
    const size_t e = blockIdx.x;  // p * N + t
    const int p = (int)(e / N);
    const int tid = threadIdx.x;
    const double2* Ae = reinterpret_cast<const double2*>(A + e * DD);
    const double2* Se = reinterpret_cast<const double2*>(S + e * DD);
    const double2* Pe = reinterpret_cast<const double2*>(Psi + e * DD);
    const double2* Ee = reinterpret_cast<const double2*>(Edf + (EDF_IMPLICIT ? 0 : e * DD));
    const double* SI = SigInv + (size_t)p * sig_stride;
    double2 pre_p[4], pre_e[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = tid + r * NT;
        const int qc = q < DD / 2 ? q : DD / 2 - 1;          // clamped: loads carry no predicate
        const int o = ((2 * qc) / D) * LD + ((2 * qc) % D);
        const double2 av = Ae[qc], sv = Se[qc];
        pre_p[r] = Pe[qc];
        if (!EDF_IMPLICIT) pre_e[r] = Ee[qc];
        if (q < DD / 2) {
            *reinterpret_cast<double2*>(Xm + o) = av;
            *reinterpret_cast<double2*>(Sm + o) = sv;
        }
    }
    if (tid < D) {
        mv[tid] = m[e * D + tid];
        lv[tid] = lam[e * D + tid];
        siv[tid] = SI[(size_t)tid * D + tid];
    }
    __syncthreads();
    if (tid < 64) {
        double s12 = 0.0;
        if (tid < D) {
            double am0 = 0.0, am1 = 0.0;
#pragma unroll 5
            for (int k = 0; k < D; k += 2) {
                const double2 av = *reinterpret_cast<const double2*>(Xm + tid * LD + k);
                am0 = fma(av.x, mv[k], am0);
                am1 = fma(av.y, mv[k + 1], am1);
            }
            const double u = -Ef[e * D + tid] - (am0 + am1) + b[e * D + tid];
            const double d = u * siv[tid];
            dbv[tid] = d;
            gLb[e * D + tid] = dt * (d + lv[tid]);
            s12 = d * mv[tid] + lv[tid] * mv[tid];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s12 += __shfl_xor(s12, off, 64);
        if (tid == 0) sc[0] = s12;
    }
    __syncthreads();   // every thread has read its share of A before X overwrites it
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = tid + r * NT;
        if (q < DD / 2) {
            const int i = (2 * q) / D, j = (2 * q) % D, o = i * LD + j;
            const double2 av = *reinterpret_cast<const double2*>(Xm + o);
            double e0, e1;
            if (EDF_IMPLICIT) {
                e0 = edf_implicit(VGPA_SYS_L96_UT, D, i, j, mv, nullptr);
                e1 = edf_implicit(VGPA_SYS_L96_UT, D, i, j + 1, mv, nullptr);
            } else {
                e0 = pre_e[r].x;
                e1 = pre_e[r].y;
            }
            const double si = siv[i];
            *reinterpret_cast<double2*>(Xm + o) =
                make_double2(si * (e0 + av.x) - 2.0 * pre_p[r].x, si * (e1 + av.y) - 2.0 * pre_p[r].y);
        }
    }
    __syncthreads();
    if (tid < NTI * NTJ) {
        const int i0 = (tid / NTJ) * GI, j0 = (tid % NTJ) * GJ;
        double acc[GI][GJ] = {};
        const double* x0 = Xm + i0 * LD;
        const double* x1 = x0 + LD;
        const double* sr = Sm + j0;
#pragma unroll 5
        for (int k = 0; k < D; k += 2) {
            const double2 a0 = *reinterpret_cast<const double2*>(x0 + k);
            const double2 a1 = *reinterpret_cast<const double2*>(x1 + k);
            const double2 p0 = *reinterpret_cast<const double2*>(sr + k * LD);
            const double2 p1 = *reinterpret_cast<const double2*>(sr + k * LD + 2);
            const double2 q0 = *reinterpret_cast<const double2*>(sr + (k + 1) * LD);
            const double2 q1 = *reinterpret_cast<const double2*>(sr + (k + 1) * LD + 2);
            const double ps[GJ] = {p0.x, p0.y, p1.x, p1.y}, qs[GJ] = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
            for (int c = 0; c < GJ; ++c) {
                acc[0][c] = fma(a0.x, ps[c], acc[0][c]);
                acc[1][c] = fma(a1.x, ps[c], acc[1][c]);
            }
#pragma unroll
            for (int c = 0; c < GJ; ++c) {
                acc[0][c] = fma(a0.y, qs[c], acc[0][c]);
                acc[1][c] = fma(a1.y, qs[c], acc[1][c]);
            }
        }
        const double s12 = sc[0];
        double out[GI][GJ];
#pragma unroll
        for (int a = 0; a < GI; ++a)
#pragma unroll
            for (int c = 0; c < GJ; ++c) {
                const double corr = (flags & VGPA_GRAD_OUTER) ? (dbv[i0 + a] * mv[j0 + c] + lv[i0 + a] * mv[j0 + c]) : s12;
                out[a][c] = dt * (acc[a][c] - corr);
            }
#pragma unroll
        for (int a = 0; a < GI; ++a) {
            double2* row = reinterpret_cast<double2*>(gLA + e * DD + (size_t)(i0 + a) * D + j0);
            row[0] = make_double2(out[a][0], out[a][1]);
            row[1] = make_double2(out[a][2], out[a][3]);
        }
    }
}

// core/gradL_Ab.py:85-106.
__global__ void grad_assemble_1d_kernel(size_t total, int N, const double* __restrict__ A, const double* __restrict__ b,
                                        const double* __restrict__ m, const double* __restrict__ S,
                                        const double* __restrict__ lam, const double* __restrict__ Psi,
                                        const double* __restrict__ Ef, const double* __restrict__ Edf,
                                        const double* __restrict__ SigInv, int sig_stride, double dt,
                                        double* __restrict__ gLA, double* __restrict__ gLb) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e / N);
    const double SI = SigInv[(size_t)p * sig_stride];
    const double db = SI * (-Ef[e] - A[e] * m[e] + b[e]);
    const double dA = SI * (Edf[e] + A[e]) * S[e] - db * m[e];
    gLA[e] = dt * (dA - lam[e] * m[e] - 2.0 * Psi[e] * S[e]);
    gLb[e] = dt * (db + lam[e]);
}

int tile_threads(int D) {
    const int ntiles = ((D + GI - 1) / GI) * ((D + GJ - 1) / GJ);
    int n = ntiles > D ? ntiles : D;
    n = vgpa::round_up(n, 64);
    return n < 64 ? 64 : n;
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int vgpa_reduce_energy(int P, int N, int C, const double* x, const double* w, double scale,
                                  double* out, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && C >= 1 && C <= 256, "vgpa_reduce_energy: bad sizes P=%d N=%d C=%d", P, N, C);
    VGPA_REQUIRE(x && w && out, "vgpa_reduce_energy: null pointer argument");
    if (P == 0) return VGPA_OK;
    hipLaunchKernelGGL(reduce_energy_kernel, dim3(P), dim3(256), 0, vgpa::as_stream(stream), N, C, x, w, scale, out);
    return vgpa::check_launch("vgpa_reduce_energy");
}

namespace {

// Zero-fill of dEobs/dm by a kernel of this library rather than hipMemsetAsync: recorded into a HIP graph,
// the runtime's memset node filled the array with other values from the second replay on (seen with ROCm 7.2
// on gfx950: every element non-zero after a replay, correct in eager mode and at the first replay).
__global__ void zero_fill_kernel(double* __restrict__ x, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = 0.0;
}

int obs_terms_impl(const char* who, int P, int N, int D, int M, int d, const double* m, const double* S,
                   const int* obs_idx, const double* obsY, int obsY_stride, const double* H, const double* Ri,
                   const double* Qi, double logconst, double* work, double* Eobs, double* dEobs_dm, double* jumpS,
                   hipStream_t st, bool packed = false) {
    {
        const size_t n = (size_t)P * N * D;
        const size_t want = (n + 255) / 256;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, dEobs_dm, n);
    }
    if (D == 1) {
        const int bs = 64, grid = (P + bs - 1) / bs;
        hipLaunchKernelGGL(obs_terms_1d_kernel, dim3(grid), dim3(bs), 0, st, P, N, M, m, S, obs_idx, obsY,
                           obsY_stride, H, Ri, logconst, Eobs, dEobs_dm, jumpS);
    } else {
        const size_t lds = sizeof(double) * 2 * D;
        if (M > 0 && (D == 2 || D == 3)) {
            const unsigned grid = (unsigned)(((size_t)P * M + 255) / 256);
            if (D == 2)
                hipLaunchKernelGGL(obs_terms_small_kernel<2>, dim3(grid), dim3(256), 0, st, P, N, M, m, S, obs_idx, obsY,
                                   obsY_stride, H, Ri, Qi, work, dEobs_dm);
            else
                hipLaunchKernelGGL(obs_terms_small_kernel<3>, dim3(grid), dim3(256), 0, st, P, N, M, m, S, obs_idx, obsY,
                                   obsY_stride, H, Ri, Qi, work, dEobs_dm);
        } else if (M > 0 && sizeof(double) * (3 * D * (D + 1) + 8 * D) <= 64 * 1024) {
            const size_t lds2 = sizeof(double) * (3 * D * (D + 1) + 8 * D);
            hipLaunchKernelGGL(obs_terms_nd_lds_kernel, dim3((M + OBS_PER_WG - 1) / OBS_PER_WG, P), dim3(256), lds2, st, N, D,
                               M, m, S, obs_idx, obsY, obsY_stride, H, Ri, Qi, work, dEobs_dm, packed ? 1 : 0);
        } else if (M > 0) {
            hipLaunchKernelGGL(obs_terms_nd_kernel, dim3(M, P), dim3(64), lds, st, N, D, M, m, S, obs_idx, obsY,
                               obsY_stride, H, Ri, Qi, work, dEobs_dm);
        }
        if (jumpS)
            hipLaunchKernelGGL(obs_jump_kernel, dim3(1), dim3(256), sizeof(double) * D * D, st, D, H, Ri, jumpS);
        hipLaunchKernelGGL(obs_finish_kernel, dim3((P + 63) / 64), dim3(64), 0, st, P, M, work, logconst, Eobs);
    }
    return vgpa::check_launch(who);
}

}  // namespace

extern "C" int vgpa_obs_terms(int P, int N, int D, int M, int d, const double* m, const double* S,
                              const int* obs_idx, const double* obsY, int obsY_stride, const double* H,
                              const double* Ri, const double* Qi, double logconst, double* work, double* Eobs,
                              double* dEobs_dm, double* jumpS, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1 && M >= 0, "vgpa_obs_terms: bad sizes P=%d N=%d D=%d M=%d", P, N, D, M);
    VGPA_REQUIRE(d == D, "vgpa_obs_terms: observation dimension d=%d must equal D=%d", d, D);
    VGPA_REQUIRE(m && S && (M == 0 || (obs_idx && obsY)) && H && Ri && Eobs && dEobs_dm, "vgpa_obs_terms: null pointer argument");
    VGPA_REQUIRE(D == 1 || (Qi && work), "vgpa_obs_terms: Qi and work are required for D > 1");
    VGPA_REQUIRE(obsY_stride == 0 || obsY_stride == M * d, "vgpa_obs_terms: obsY_stride must be 0 or M*d");
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_obs_terms: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    if (P == 0) return VGPA_OK;
    return obs_terms_impl("vgpa_obs_terms", P, N, D, M, d, m, S, obs_idx, obsY, obsY_stride, H, Ri, Qi, logconst, work,
                          Eobs, dEobs_dm, jumpS, vgpa::as_stream(stream));
}

// vgpa_obs_terms with S in the packed symmetric layout (D = 40): the kernel reads the diagonal of S only.
extern "C" int vgpa_obs_terms_packed(int P, int N, int M, int d, const double* m, const double* S_packed,
                                     const int* obs_idx, const double* obsY, int obsY_stride, const double* H,
                                     const double* Ri, const double* Qi, double logconst, double* work, double* Eobs,
                                     double* dEobs_dm, double* jumpS, void* stream) {
    constexpr int D = 40;
    VGPA_REQUIRE(P >= 0 && N >= 1 && M >= 0, "vgpa_obs_terms_packed: bad sizes P=%d N=%d M=%d", P, N, M);
    VGPA_REQUIRE(d == D, "vgpa_obs_terms_packed: observation dimension d=%d must equal D=40", d);
    VGPA_REQUIRE(m && S_packed && (M == 0 || (obs_idx && obsY)) && H && Ri && Qi && work && Eobs && dEobs_dm,
                 "vgpa_obs_terms_packed: null pointer argument");
    VGPA_REQUIRE(obsY_stride == 0 || obsY_stride == M * d, "vgpa_obs_terms_packed: obsY_stride must be 0 or M*d");
    if (P == 0) return VGPA_OK;
    return obs_terms_impl("vgpa_obs_terms_packed", P, N, D, M, d, m, S_packed, obs_idx, obsY, obsY_stride, H, Ri, Qi,
                          logconst, work, Eobs, dEobs_dm, jumpS, vgpa::as_stream(stream), true);
}

// ---------------------------------------------------------------------------
// The observation terms for callers that cannot allocate: the per-observation energies
// (P * M doubles, what vgpa_obs_terms takes as `work`) go to a fixed array of the code object.
// ---------------------------------------------------------------------------
static __device__ __align__(256) double scratch[VGPA_OBS_STATIC_MAX];

namespace {
constexpr int kMaxDevices = 64;
// Device address of `scratch` on each GPU this process has used (every GPU holds its own copy
// of the array at its own address); resolved once per device, with that device current.
double* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

int scratch_for_current_device(hipStream_t st, double** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        (void)hipGetLastError();
        return vgpa::fail(VGPA_ELAUNCH, "vgpa_obs_terms_static: cannot identify the current device");
    }
    // The array is looked up for the CURRENT device; a stream of another GPU would run the kernels
    // there with this GPU's address.  (The null stream belongs to the current device.)
    if (st) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(st, &sdev) != hipSuccess) {
            (void)hipGetLastError();
        } else if ((int)sdev != dev) {
            return vgpa::fail(VGPA_EINVAL, "vgpa_obs_terms_static: the stream belongs to device %d but device %d is "
                              "current; make the stream's device current for this call", (int)sdev, dev);
        }
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            (void)hipGetLastError();
            return vgpa::fail(VGPA_ELAUNCH, "vgpa_obs_terms_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        }
        g_scratch_ptr[dev] = static_cast<double*>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return VGPA_OK;
}
}  // namespace

extern "C" int vgpa_obs_terms_static(int P, int N, int D, int M, int d, const double* m, const double* S,
                                     const int* obs_idx, const double* obsY, int obsY_stride, const double* H,
                                     const double* Ri, const double* Qi, double logconst, double* Eobs,
                                     double* dEobs_dm, double* jumpS, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1 && M >= 0, "vgpa_obs_terms_static: bad sizes P=%d N=%d D=%d M=%d", P, N, D, M);
    VGPA_REQUIRE(d == D, "vgpa_obs_terms_static: observation dimension d=%d must equal D=%d", d, D);
    VGPA_REQUIRE(m && S && (M == 0 || (obs_idx && obsY)) && H && Ri && Eobs && dEobs_dm,
                 "vgpa_obs_terms_static: null pointer argument");
    VGPA_REQUIRE(D == 1 || Qi, "vgpa_obs_terms_static: Qi is required for D > 1");
    VGPA_REQUIRE(obsY_stride == 0 || obsY_stride == M * d, "vgpa_obs_terms_static: obsY_stride must be 0 or M*d");
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_obs_terms_static: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    if (D > 1 && (size_t)P * (size_t)M > (size_t)VGPA_OBS_STATIC_MAX)
        return vgpa::fail(VGPA_EINVAL, "vgpa_obs_terms_static: P*M = %zu exceeds the fixed scratch of %d doubles "
                          "(VGPA_OBS_STATIC_MAX); use vgpa_obs_terms with a caller-provided work array",
                          (size_t)P * (size_t)M, (int)VGPA_OBS_STATIC_MAX);
    if (P == 0) return VGPA_OK;
    double* work = nullptr;
    const int rc = scratch_for_current_device(vgpa::as_stream(stream), &work);
    if (rc != VGPA_OK) return rc;
    // every slot obs_finish_kernel reads (p * M + n, n < M) is written by obs_terms_nd_kernel of
    // this call first, so nothing left by an earlier call is ever read
    return obs_terms_impl("vgpa_obs_terms_static", P, N, D, M, d, m, S, obs_idx, obsY, obsY_stride, H, Ri, Qi, logconst,
                          work, Eobs, dEobs_dm, jumpS, vgpa::as_stream(stream));
}

extern "C" int vgpa_kl0(int P, int D, const double* m0, const double* S0, const double* mu0, int mu0_stride,
                        const double* tau0, int tau0_stride, int m0_is_row, double* KL0, void* stream) {
    VGPA_REQUIRE(P >= 0 && D >= 1, "vgpa_kl0: bad sizes P=%d D=%d", P, D);
    VGPA_REQUIRE(m0 && S0 && mu0 && tau0 && KL0, "vgpa_kl0: null pointer argument");
    VGPA_REQUIRE(mu0_stride == 0 || mu0_stride == D, "vgpa_kl0: mu0_stride must be 0 or D");
    VGPA_REQUIRE(tau0_stride == 0 || tau0_stride == D * D, "vgpa_kl0: tau0_stride must be 0 or D*D");
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_kl0: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    if (P == 0) return VGPA_OK;
    hipStream_t st = vgpa::as_stream(stream);
    if (D == 1) {
        const int bs = 64, grid = (P + bs - 1) / bs;
        hipLaunchKernelGGL(kl0_1d_kernel, dim3(grid), dim3(bs), 0, st, P, m0, S0, mu0, mu0_stride, tau0, tau0_stride, KL0);
    } else {
        const size_t lds = sizeof(double) * ((size_t)4 * D * (D + 1) + D);
        allow_lds(kl0_nd_kernel, lds);
        hipLaunchKernelGGL(kl0_nd_kernel, dim3(P), dim3(64), lds, st, D, m0, S0, mu0, mu0_stride, tau0, tau0_stride,
                           m0_is_row, KL0);
    }
    return vgpa::check_launch("vgpa_kl0");
}

extern "C" int vgpa_grad_assemble(int P, int N, int D, const double* A, const double* b, const double* m,
                                  const double* S, const double* lam, const double* Psi, const double* Ef,
                                  const double* Edf, int edf_system, const double* theta, int theta_stride,
                                  const double* SigInv, int sig_stride, double dt, int flags,
                                  double* gLA, double* gLb, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_grad_assemble: bad sizes P=%d N=%d D=%d", P, N, D);
    VGPA_REQUIRE(A && b && m && S && lam && Psi && Ef && SigInv && gLA && gLb, "vgpa_grad_assemble: null pointer argument");
    VGPA_REQUIRE(sig_stride == 0 || sig_stride == D * D, "vgpa_grad_assemble: sig_stride must be 0 or D*D");
    VGPA_REQUIRE((size_t)P * N < 0x7fffffffull, "vgpa_grad_assemble: P*N too large");
    if (!Edf) {
        VGPA_REQUIRE(D > 1, "vgpa_grad_assemble: Edf is required for D == 1");
        VGPA_REQUIRE(edf_system == VGPA_SYS_L96_UT || ((edf_system == VGPA_SYS_L63 || edf_system == VGPA_SYS_L63_UT) && D == 3 && theta),
                     "vgpa_grad_assemble: Edf is NULL and edf_system=%d cannot be derived from the mean", edf_system);
    }
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_grad_assemble: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    const size_t total = (size_t)P * N;
    if (total == 0) return VGPA_OK;
    hipStream_t st = vgpa::as_stream(stream);
    if (D == 1) {
        const int bs = 256;
        hipLaunchKernelGGL(grad_assemble_1d_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st, total, N,
                           A, b, m, S, lam, Psi, Ef, Edf, SigInv, sig_stride, dt, gLA, gLb);
    } else {
        const int sig_diag = (flags & VGPA_GRAD_SIGINV_DIAG) ? 1 : 0;
        const bool implicit96 = !Edf && edf_system == VGPA_SYS_L96_UT;
        if (sig_diag && (D == 40 || D == 16) && (Edf || implicit96)) {
            const size_t lds = sizeof(double) * ((size_t)2 * D * (D + 2) + 4 * D + 2);
#define VGPA_LAUNCH_GT(DD_, IMP)                                                                              \
    do {                                                                                                      \
        allow_lds(grad_assemble_tiled_kernel<DD_, IMP>, lds);                                                 \
        hipLaunchKernelGGL((grad_assemble_tiled_kernel<DD_, IMP>), dim3((unsigned)total), dim3(256), lds, st, N, A, b, \
                           m, S, lam, Psi, Ef, Edf ? Edf : A, SigInv, sig_stride, dt, flags, gLA, gLb);      \
    } while (0)
            if (D == 40) { if (implicit96) VGPA_LAUNCH_GT(40, true); else VGPA_LAUNCH_GT(40, false); }
            else { if (implicit96) VGPA_LAUNCH_GT(16, true); else VGPA_LAUNCH_GT(16, false); }
#undef VGPA_LAUNCH_GT
        } else if (D >= 2 && D <= 4) {
            const unsigned grid = (unsigned)((total + 255) / 256);
#define VGPA_LAUNCH_GS(DD_)                                                                                         \
    hipLaunchKernelGGL(grad_assemble_small_kernel<DD_>, dim3(grid), dim3(256), 0, st, total, N, A, b, m, S, lam, Psi, \
                       Ef, Edf, edf_system, theta, theta_stride, SigInv, sig_stride, sig_diag, dt, flags, gLA, gLb)
            if (D == 2) VGPA_LAUNCH_GS(2); else if (D == 3) VGPA_LAUNCH_GS(3); else VGPA_LAUNCH_GS(4);
#undef VGPA_LAUNCH_GS
        } else {
            const size_t lds = sizeof(double) * ((size_t)3 * D * (D + 1) + 4 * D + 2);
            allow_lds(grad_assemble_nd_kernel, lds);
            hipLaunchKernelGGL(grad_assemble_nd_kernel, dim3((unsigned)total), dim3(tile_threads(D)), lds, st, N, D, A,
                               b, m, S, lam, Psi, Ef, Edf, edf_system, theta, theta_stride, SigInv, sig_stride,
                               sig_diag, dt, flags, gLA, gLb);
        }
    }
    return vgpa::check_launch("vgpa_grad_assemble");
}
