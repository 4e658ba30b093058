// Stand-alone pieces of the unscented machinery, for callers that drive the
// transform themselves with their own function f (the per-step energy kernels have
// all of this fused in): sigma points, the weighted moments of the transformed
// points, the per-sample gradient terms, and the Gaussian moment table.
//
//   vgpa_ut_sigma_points  auxiliary/numerical.py:186-213
//   vgpa_ut_moments       auxiliary/numerical.py:216-226
//   vgpa_grad_esde_ms     core/gradEsde_mS.py:8-68
//   vgpa_mom_gauss        auxiliary/momGauss.py:5-107
#include "common.h"
#include "linalg.h"

#include <cmath>

namespace {

// One wavefront per problem.  LDS: D (D+1) doubles.
__global__ void __launch_bounds__(64)
ut_sigma_points_kernel(int D, const double* __restrict__ xbar, const double* __restrict__ Pxx,
                       double* __restrict__ chi, int* __restrict__ status) {
    extern __shared__ double lds[];
    const int LD = D + 1, l = threadIdx.x, p = blockIdx.x;
    const size_t DD = (size_t)D * D;
    double* Lm = lds;
    const double* Pg = Pxx + (size_t)p * DD;
    const double kappa = 1.05 * D;          // numerical.py:196
    const double c2 = D + kappa;
    for (int q = l; q < (int)DD; q += 64) Lm[(q / D) * LD + (q % D)] = c2 * Pg[q];   // numerical.py:200
    const bool act = l < D;
    __syncthreads();
    bool ok = true;
    for (int j = 0; j < D; ++j) {
        const double djj = Lm[j * LD + j];
        if (!(djj > 0.0)) { ok = false; break; }
        const double d = sqrt(djj);
        double lij = 0.0;
        if (act && l >= j) {
            lij = (l == j) ? d : Lm[l * LD + j] / d;
            Lm[l * LD + j] = lij;
        }
        __syncthreads();
        for (int k = j + 1; k < D; ++k) {
            const double lkj = Lm[k * LD + j];
            if (act && l >= k) Lm[l * LD + k] = fma(-lij, lkj, Lm[l * LD + k]);
        }
        __syncthreads();
    }
    int st = 0;
    if (!ok) {
        // diagonal, unscaled (numerical.py:202)
        st = VGPA_ST_CHOL_FALLBACK;
        __syncthreads();
        const double dg = act ? Pg[(size_t)l * D + l] : 1.0;
        if (!__all(dg > 0.0)) st |= VGPA_ST_CHOL_FAILED;
        for (int q = l; q < D * LD; q += 64) Lm[q] = 0.0;
        __syncthreads();
        if (act) Lm[l * LD + l] = (st & VGPA_ST_CHOL_FAILED) ? __builtin_nan("") : sqrt(dg);
        __syncthreads();
    }
    if (l == 0 && status) status[p] = st;
    // chi[0] = xbar; chi[1+i] = xbar + column i of L; chi[1+D+i] = xbar - column i of L
    const int K = 2 * D + 1;
    double* out = chi + (size_t)p * K * D;
    const double* xb = xbar + (size_t)p * D;
    for (int q = l; q < K * D; q += 64) {
        const int k = q / D, j = q % D;
        double v = xb[j];
        if (k >= 1) {
            const int i = (k - 1) % D;
            const double u = (j >= i) ? Lm[j * LD + i] : 0.0;
            v = (k <= D) ? v + u : v - u;
        }
        out[q] = v;
    }
}

// ybar[c] = sum_k w_k Y[k][c]
__global__ void ut_mean_kernel(int P, int K, int C, const double* __restrict__ Y, double w0, double w1,
                               double* __restrict__ ybar) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)P * C) return;
    const int p = (int)(idx / C), c = (int)(idx % C);
    const double* y = Y + (size_t)p * K * C + c;
    double s = w0 * y[0];
    for (int k = 1; k < K; ++k) s = fma(w1, y[(size_t)k * C], s);
    ybar[idx] = s;
}

// Pyy = Y' Q Y with Q = (I - 1 w') diag(w) (I - 1 w')'  (numerical.py:222-226).
// Entry (i,l) of Q is  delta_il w_i - w_i^2 - w_l^2 + sum_k w_k^3,  so with
// q = sum_k w_k^2 y_k and t = sum_k y_k:
//   Pyy = sum_k w_k y_k y_k' - q t' - t q' + (sum_k w_k^3) t t'.
// 16 x 16 outputs per block.
__global__ void __launch_bounds__(256)
ut_cov_kernel(int K, int C, const double* __restrict__ Y, double w0, double w1, double w3,
              double* __restrict__ Pyy) {
    const int p = blockIdx.z;
    const int a = blockIdx.y * 16 + threadIdx.y, c = blockIdx.x * 16 + threadIdx.x;
    if (a >= C || c >= C) return;
    const double* y = Y + (size_t)p * K * C;
    double sac = 0.0, qa = 0.0, qc = 0.0, ta = 0.0, tc = 0.0;
    for (int k = 0; k < K; ++k) {
        const double w = k == 0 ? w0 : w1;
        const double ya = y[(size_t)k * C + a], yc = y[(size_t)k * C + c];
        sac = fma(w * ya, yc, sac);
        qa = fma(w * w, ya, qa);
        qc = fma(w * w, yc, qc);
        ta += ya;
        tc += yc;
    }
    Pyy[(size_t)p * C * C + (size_t)a * C + c] = sac - qa * tc - ta * qc + w3 * ta * tc;
}

// One wavefront per sample k.  LDS: 2 D (D+1) + 4 D doubles.
__global__ void __launch_bounds__(64)
grad_esde_ms_kernel(int D, const double* __restrict__ x, const double* __restrict__ Fx,
                    const double* __restrict__ mt, const double* __restrict__ St,
                    const double* __restrict__ At, const double* __restrict__ bt,
                    const double* __restrict__ dSigI, double* __restrict__ out) {
    extern __shared__ double lds[];
    const int LD = D + 1, MS = D * LD, l = threadIdx.x, k = blockIdx.x;
    double* Sm = lds;
    double* Si = lds + MS;
    double* xv = lds + 2 * MS;
    double* zv = xv + D;
    double* uv = zv + D;     // Si z
    double* rv = uv + D;     // z' Si
    for (int q = l; q < D * D; q += 64) Sm[(q / D) * LD + (q % D)] = St[q];
    const bool act = l < D;
    double xl = 0.0;
    if (act) {
        xl = x[(size_t)k * D + l];
        xv[l] = xl;
        zv[l] = xl - mt[l];
    }
    __syncthreads();
    double r2 = 0.0;
    if (act) {
        double ax = 0.0;
        for (int j = 0; j < D; ++j) ax = fma(At[(size_t)l * D + j], xv[j], ax);
        const double r = Fx[(size_t)k * D + l] + ax - bt[l];          // gradEsde_mS.py:44
        r2 = dSigI[l] * r * r;
    }
    double v = r2;                                                    // :45
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    vgpa::gj_inverse(Sm, Si, D, LD, l);
    double* o = out + (size_t)k * D * (D + 1);
    if (act) {
        double sx = 0.0, sz = 0.0, zs = 0.0;
        for (int j = 0; j < D; ++j) {
            sx = fma(Si[l * LD + j], xv[j], sx);
            sz = fma(Si[l * LD + j], zv[j], sz);
            zs = fma(zv[j], Si[j * LD + l], zs);
        }
        o[l] = 0.5 * v * sx;                                          // :48, :63
        uv[l] = sz;
        rv[l] = zs;
    }
    __syncthreads();
    for (int q = l; q < D * D; q += 64) o[D + q] = 0.5 * v * uv[q / D] * rv[q % D];   // :60, :64
}

// auxiliary/momGauss.py:37-101.  deriv: 0 = moment, 1 = d/dm, 2 = d/dS.
__device__ inline double moment(double m, double S, int order) {
    const double m2 = m * m, S2 = S * S;
    switch (order) {
        case 0: return 1.0;
        case 1: return m;
        case 2: return m2 + S;
        case 3: return m2 * m + 3 * m * S;
        case 4: return m2 * m2 + 6 * m2 * S + 3 * S2;
        case 5: return m2 * m2 * m + 10 * (m2 * m) * S + 15 * m * S2;
        case 6: return m2 * m2 * m2 + 15 * (m2 * m2) * S + 45 * m2 * S2 + 15 * (S2 * S);
        case 7: return m2 * m2 * m2 * m + 21 * (m2 * m2 * m) * S + 105 * (m2 * m) * S2 + 105 * m * (S2 * S);
        default: return m2 * m2 * m2 * m2 + 28 * (m2 * m2 * m2) * S + 210 * (m2 * m2) * S2 + 420 * m2 * (S2 * S) + 105 * (S2 * S2);
    }
}

__global__ void mom_gauss_kernel(size_t n, const double* __restrict__ m, const double* __restrict__ S,
                                 int order, int deriv, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mi = m[i], Si = S[i];
    double v;
    if (deriv == 0) {
        v = moment(mi, Si, order);
    } else if (deriv == 1) {
        v = order == 0 ? 1.0 : order * moment(mi, Si, order - 1);     // d<x^n>/dm = n <x^(n-1)>
        if (order == 0) v = 1.0;
    } else {
        const double m2 = mi * mi, S2 = Si * Si;
        switch (order) {
            case 0: v = 1.0; break;
            case 1: v = 0.0; break;
            case 2: v = 1.0; break;
            case 3: v = 3 * mi; break;
            case 4: v = 6 * (m2 + Si); break;
            case 5: v = 10 * (m2 * mi) + 30 * (mi * Si); break;
            case 6: v = 15 * (m2 * m2) + 90 * m2 * Si + 45 * S2; break;
            case 7: v = 21 * (m2 * m2 * mi) + 210 * (m2 * mi) * Si + 315 * mi * S2; break;
            default: v = 28 * (m2 * m2 * m2) + 420 * (m2 * m2) * Si + 1260 * m2 * S2 + 420 * (S2 * Si); break;
        }
    }
    out[i] = v;
}

void allow_lds(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int vgpa_ut_sigma_points(int P, int D, const double* xbar, const double* Pxx, double* chi,
                                    int* status, void* stream) {
    VGPA_REQUIRE(P >= 0 && D >= 1, "vgpa_ut_sigma_points: bad sizes P=%d D=%d", P, D);
    VGPA_REQUIRE(xbar && Pxx && chi, "vgpa_ut_sigma_points: null pointer argument");
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_ut_sigma_points: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    if (P == 0) return VGPA_OK;
    const size_t lds = sizeof(double) * (size_t)D * (D + 1);
    hipLaunchKernelGGL(ut_sigma_points_kernel, dim3(P), dim3(64), lds, vgpa::as_stream(stream), D, xbar, Pxx, chi, status);
    return vgpa::check_launch("vgpa_ut_sigma_points");
}

extern "C" int vgpa_ut_moments(int P, int K, int C, const double* Y, double* ybar, double* Pyy, void* stream) {
    VGPA_REQUIRE(P >= 0 && K >= 3 && (K & 1) && C >= 1, "vgpa_ut_moments: bad sizes P=%d K=%d C=%d (K = 2D+1)", P, K, C);
    VGPA_REQUIRE(Y && ybar, "vgpa_ut_moments: null pointer argument");
    VGPA_REQUIRE(P <= 65535, "vgpa_ut_moments: P=%d too large", P);
    if (P == 0) return VGPA_OK;
    const int D = (K - 1) / 2;
    const double kappa = 1.05 * D, c2 = D + kappa;
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);             // numerical.py:211-212
    const double w3 = w0 * w0 * w0 + (K - 1) * (w1 * w1 * w1);
    hipStream_t st = vgpa::as_stream(stream);
    const size_t total = (size_t)P * C;
    hipLaunchKernelGGL(ut_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, K, C, Y, w0, w1, ybar);
    if (Pyy) {
        const unsigned g = (unsigned)((C + 15) / 16);
        hipLaunchKernelGGL(ut_cov_kernel, dim3(g, g, P), dim3(16, 16), 0, st, K, C, Y, w0, w1, w3, Pyy);
    }
    return vgpa::check_launch("vgpa_ut_moments");
}

extern "C" int vgpa_grad_esde_ms(int K, int D, const double* x, const double* Fx, const double* mt, const double* St,
                                 const double* At, const double* bt, const double* dSigI, double* out, void* stream) {
    VGPA_REQUIRE(K >= 0 && D >= 1, "vgpa_grad_esde_ms: bad sizes K=%d D=%d", K, D);
    VGPA_REQUIRE(x && Fx && mt && St && At && bt && dSigI && out, "vgpa_grad_esde_ms: null pointer argument");
    if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_grad_esde_ms: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
    if (K == 0) return VGPA_OK;
    const size_t lds = sizeof(double) * ((size_t)2 * D * (D + 1) + 4 * D);
    allow_lds(reinterpret_cast<const void*>(grad_esde_ms_kernel), lds);
    hipLaunchKernelGGL(grad_esde_ms_kernel, dim3(K), dim3(64), lds, vgpa::as_stream(stream), D, x, Fx, mt, St, At, bt, dSigI, out);
    return vgpa::check_launch("vgpa_grad_esde_ms");
}

extern "C" int vgpa_mom_gauss(long long n, const double* m, const double* S, int order, int deriv, double* out,
                              void* stream) {
    VGPA_REQUIRE(n >= 0, "vgpa_mom_gauss: bad size n=%lld", n);
    VGPA_REQUIRE(order >= 0 && order <= 8, "vgpa_mom_gauss: Unknown Higher Order Gaussian Moment! (order=%d)", order);
    VGPA_REQUIRE(deriv >= 0 && deriv <= 2, "vgpa_mom_gauss: deriv=%d must be 0 (moment), 1 (d/dm) or 2 (d/dS)", deriv);
    VGPA_REQUIRE(m && S && out, "vgpa_mom_gauss: null pointer argument");
    if (n == 0) return VGPA_OK;
    hipLaunchKernelGGL(mom_gauss_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vgpa::as_stream(stream),
                       (size_t)n, m, S, order, deriv, out);
    return vgpa::check_launch("vgpa_mom_gauss");
}
