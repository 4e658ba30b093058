// Small state dimensions (D = 2, 3, i.e. the Lorenz-63 grid searches): one lane
// per problem for the scans and one lane per (problem, step) for the gradient
// assembly, everything in registers.  A workgroup per problem would leave all but a
// few lanes idle at these sizes and pay a barrier per stage.
//
// The scans are sequential in time, so a lane hides the latency of its own loads by
// keeping the inputs of the next RING - 1 steps in flight (a ring of register sets,
// the step loop unrolled RING times so that every ring index is static).
//
// Included inside the anonymous namespace of scan.hip (scans) and assemble.hip
// (gradient assembly).
#pragma once

template <int D>
struct SmallD {
    static constexpr int DD = D * D;
    // rows kept in flight per lane, as many as the 256 vector registers allow
    static constexpr int ring_fwd(int method) { return method == VGPA_RK4 ? 6 : 8; }
    static constexpr int ring_bwd(int method) { return method == VGPA_RK4 ? 4 : 6; }

    // K = Sig - A X - X A'   (core/fwd_ode.py:45)
    static __device__ __forceinline__ void fwd_rhs(const double (&A)[DD], const double (&X)[DD], const double (&Sg)[DD],
                                                   double (&K)[DD]) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    s1 = fma(A[i * D + k], X[k * D + j], s1);
                    s2 = fma(X[i * D + k], A[j * D + k], s2);
                }
                K[i * D + j] = -s1 - s2 + Sg[i * D + j];
            }
    }
    // f = -A x + b   (core/fwd_ode.py:44)
    static __device__ __forceinline__ void fwd_rhs_vec(const double (&A)[DD], const double (&x)[D], const double (&b)[D],
                                                       double (&f)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s = fma(A[i * D + k], x[k], s);
            f[i] = -s + b[i];
        }
    }
    // K = -G + X A + A' X   (core/bwd_ode.py:43)
    static __device__ __forceinline__ void bwd_rhs(const double (&A)[DD], const double (&X)[DD], const double (&G)[DD],
                                                   double (&K)[DD]) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    s1 = fma(X[i * D + k], A[k * D + j], s1);
                    s2 = fma(A[k * D + i], X[k * D + j], s2);
                }
                K[i * D + j] = -G[i * D + j] + (s1 + s2);
            }
    }
    // f = -g + A x   (core/bwd_ode.py:42: lam A')
    static __device__ __forceinline__ void bwd_rhs_vec(const double (&A)[DD], const double (&x)[D], const double (&g)[D],
                                                       double (&f)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s = fma(A[i * D + k], x[k], s);
            f[i] = -g[i] + s;
        }
    }
    template <int n>
    static __device__ __forceinline__ void load(const double* __restrict__ src, double (&dst)[n]) {
#pragma unroll
        for (int q = 0; q < n; ++q) dst[q] = src[q];
    }
    template <int n>
    static __device__ __forceinline__ void store(double* __restrict__ dst, const double (&src)[n]) {
#pragma unroll
        for (int q = 0; q < n; ++q) dst[q] = src[q];
    }
    template <int n>
    static __device__ __forceinline__ void mid(const double (&a)[n], const double (&b)[n], double (&m)[n]) {
#pragma unroll
        for (int q = 0; q < n; ++q) m[q] = 0.5 * (a[q] + b[q]);
    }
    // y = x + c k
    template <int n>
    static __device__ __forceinline__ void axpy(const double (&x)[n], double c, const double (&k)[n], double (&y)[n]) {
#pragma unroll
        for (int q = 0; q < n; ++q) y[q] = x[q] + c * k[q];
    }
};

#ifdef VGPA_SMALL_D_SCANS
// ---------------------------------------------------------------------------
// forward scan: core/fwd_ode.py euler_nD :87, heun_nD :181, rk2_nD :311, rk4_nD :426
// ---------------------------------------------------------------------------
template <int D, int METHOD>
__global__ void __launch_bounds__(64, 1)
fwd_scan_small_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ b,
                      const double* __restrict__ m0, const double* __restrict__ S0,
                      const double* __restrict__ Sigma, int sigma_stride, double dt,
                      double* __restrict__ m_out, double* __restrict__ S_out) {
    using T = SmallD<D>;
    constexpr int DD = T::DD, RING = T::ring_fwd(METHOD);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double* Ap = A + (size_t)p * N * DD;
    const double* bp = b + (size_t)p * N * D;
    double* mo = m_out + (size_t)p * N * D;
    double* So = S_out + (size_t)p * N * DD;
    const double h = 0.5 * dt;
    double Sg[DD], S[DD], m[D];
    T::load(Sigma + (size_t)p * sigma_stride, Sg);
    T::load(S0 + (size_t)p * DD, S);
    T::load(m0 + (size_t)p * D, m);
    T::store(So, S);
    T::store(mo, m);
    double Ar[RING][DD], br[RING][D];   // slot u holds row t with t % RING == u
#pragma unroll
    for (int u = 0; u < RING; ++u) {
        const int r = u < N ? u : N - 1;
        T::load(Ap + (size_t)r * DD, Ar[u]);
        T::load(bp + (size_t)r * D, br[u]);
    }
    for (int t0 = 0; t0 + 1 < N; t0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int t = t0 + u;
            if (t + 1 >= N) break;
            constexpr int dummy = 0; (void)dummy;
            const double (&At)[DD] = Ar[u];
            const double (&bt)[D] = br[u];
            const double (&An)[DD] = Ar[(u + 1) % RING];
            const double (&bn)[D] = br[(u + 1) % RING];
            double k1[DD], f1[D], Sn[DD], mn[D];
            T::fwd_rhs(At, S, Sg, k1);
            T::fwd_rhs_vec(At, m, bt, f1);
            if (METHOD == VGPA_EULER) {
                T::axpy(S, dt, k1, Sn);
                T::axpy(m, dt, f1, mn);
            } else if (METHOD == VGPA_HEUN) {
                double tS[DD], tm[D], k2[DD], f2[D];
                T::axpy(S, dt, k1, tS);
                T::axpy(m, dt, f1, tm);
                T::fwd_rhs(An, tS, Sg, k2);
                T::fwd_rhs_vec(An, tm, bn, f2);
#pragma unroll
                for (int q = 0; q < DD; ++q) Sn[q] = S[q] + h * (k1[q] + k2[q]);
#pragma unroll
                for (int q = 0; q < D; ++q) mn[q] = m[q] + h * (f1[q] + f2[q]);
            } else if (METHOD == VGPA_RK2) {
                double Ak[DD], bk[D], tS[DD], tm[D], k2[DD], f2[D];
                T::mid(At, An, Ak);
                T::mid(bt, bn, bk);
                T::axpy(S, h, k1, tS);
                T::axpy(m, h, f1, tm);
                T::fwd_rhs(Ak, tS, Sg, k2);
                T::fwd_rhs_vec(Ak, tm, bk, f2);
                T::axpy(S, dt, k2, Sn);
                T::axpy(m, dt, f2, mn);
            } else {
                double Ak[DD], bk[D], tS[DD], tm[D], L1[DD], L2[DD], L3[DD], L4[DD], K1[D], K2[D], K3[D], K4[D];
                T::mid(At, An, Ak);
                T::mid(bt, bn, bk);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L1[q] = k1[q] * dt; tS[q] = S[q] + 0.5 * L1[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K1[q] = f1[q] * dt; tm[q] = m[q] + 0.5 * K1[q]; }
                T::fwd_rhs(Ak, tS, Sg, L2);
                T::fwd_rhs_vec(Ak, tm, bk, K2);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L2[q] *= dt; tS[q] = S[q] + 0.5 * L2[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K2[q] *= dt; tm[q] = m[q] + 0.5 * K2[q]; }
                T::fwd_rhs(Ak, tS, Sg, L3);
                T::fwd_rhs_vec(Ak, tm, bk, K3);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L3[q] *= dt; tS[q] = S[q] + L3[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K3[q] *= dt; tm[q] = m[q] + K3[q]; }
                T::fwd_rhs(An, tS, Sg, L4);
                T::fwd_rhs_vec(An, tm, bn, K4);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L4[q] *= dt; Sn[q] = S[q] + (L1[q] + 2.0 * (L2[q] + L3[q]) + L4[q]) / 6.0; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K4[q] *= dt; mn[q] = m[q] + (K1[q] + 2.0 * (K2[q] + K3[q]) + K4[q]) / 6.0; }
            }
#pragma unroll
            for (int q = 0; q < DD; ++q) S[q] = Sn[q];
#pragma unroll
            for (int q = 0; q < D; ++q) m[q] = mn[q];
            T::store(So + (size_t)(t + 1) * DD, S);
            T::store(mo + (size_t)(t + 1) * D, m);
            // slot u is free: fetch row t + RING (clamped: a redundant load instead of a branch)
            const int r = t + RING < N ? t + RING : N - 1;
            T::load(Ap + (size_t)r * DD, Ar[u]);
            T::load(bp + (size_t)r * D, br[u]);
        }
    }
}

// ---------------------------------------------------------------------------
// backward scan: core/bwd_ode.py euler_nD :86, heun_nD :177, rk2_nD :305, rk4_nD :427
// ---------------------------------------------------------------------------
template <int D, int METHOD>
__global__ void __launch_bounds__(64, 1)
bwd_scan_small_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                      const double* __restrict__ dEdS, const double* __restrict__ jm,
                      const double* __restrict__ jS_dense, const int* __restrict__ obs_flag,
                      const double* __restrict__ jumpS, int jumpS_stride, double dt,
                      const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                      double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    using T = SmallD<D>;
    constexpr int DD = T::DD, RING = T::ring_bwd(METHOD);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double* Ap = A + (size_t)p * N * DD;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * DD;
    const double* jmp = jm + (size_t)p * N * D;
    const double* jSd = jS_dense ? jS_dense + (size_t)p * N * DD : nullptr;
    double* lo = lam_out + (size_t)p * N * D;
    double* Po = Psi_out + (size_t)p * N * DD;
    const double h = 0.5 * dt;
    double jt[DD], X[DD], x[D];
    const bool flagged = !jSd && jumpS && obs_flag;
#pragma unroll
    for (int q = 0; q < DD; ++q) {
        jt[q] = flagged ? jumpS[(size_t)p * jumpS_stride + q] : 0.0;
        X[q] = Psi_last ? Psi_last[(size_t)p * DD + q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < D; ++q) x[q] = lam_last ? lam_last[(size_t)p * D + q] : 0.0;
    T::store(Po + (size_t)(N - 1) * DD, X);
    T::store(lo + (size_t)(N - 1) * D, x);
    // slot u holds row t = N - 1 - s with s % RING == u
    double Ar[RING][DD], Gr[RING][DD], gr[RING][D], jr[RING][D];
    int fr[RING];                                   // observation flag of the row
#pragma unroll
    for (int u = 0; u < RING; ++u) {
        const int r = N - 1 - u > 0 ? N - 1 - u : 0;
        T::load(Ap + (size_t)r * DD, Ar[u]);
        T::load(gS + (size_t)r * DD, Gr[u]);
        T::load(gm + (size_t)r * D, gr[u]);
        T::load(jmp + (size_t)r * D, jr[u]);
        fr[u] = flagged ? obs_flag[r] : 0;
    }
    for (int s0 = 0; s0 + 1 < N; s0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int t = N - 1 - (s0 + u);
            if (t < 1) break;
            const double (&At)[DD] = Ar[u];
            const double (&Gt)[DD] = Gr[u];
            const double (&gt)[D] = gr[u];
            const double (&Ap1)[DD] = Ar[(u + 1) % RING];   // row t - 1
            const double (&Gp)[DD] = Gr[(u + 1) % RING];
            const double (&gp)[D] = gr[(u + 1) % RING];
            // jumps at t - 1 (core/bwd_ode.py:124,127)
            const double (&jv)[D] = jr[(u + 1) % RING];
            double jd[DD];
            if (jSd) {
                T::load(jSd + (size_t)(t - 1) * DD, jd);
            } else {
                const bool on = fr[(u + 1) % RING] != 0;
#pragma unroll
                for (int q = 0; q < DD; ++q) jd[q] = on ? jt[q] : 0.0;
            }
            double k1[DD], f1[D], Xn[DD], xn[D];
            T::bwd_rhs(At, X, Gt, k1);
            T::bwd_rhs_vec(At, x, gt, f1);
            if (METHOD == VGPA_EULER) {
                T::axpy(X, -dt, k1, Xn);
                T::axpy(x, -dt, f1, xn);
            } else if (METHOD == VGPA_HEUN) {
                double tX[DD], tx[D], k2[DD], f2[D];
                T::axpy(X, -dt, k1, tX);
                T::axpy(x, -dt, f1, tx);
                T::bwd_rhs(Ap1, tX, Gp, k2);
                T::bwd_rhs_vec(Ap1, tx, gp, f2);
#pragma unroll
                for (int q = 0; q < DD; ++q) Xn[q] = X[q] - h * (k1[q] + k2[q]);
#pragma unroll
                for (int q = 0; q < D; ++q) xn[q] = x[q] - h * (f1[q] + f2[q]);
            } else if (METHOD == VGPA_RK2) {
                double Ak[DD], Gk[DD], gk[D], tX[DD], tx[D], k2[DD], f2[D];
                T::mid(Ap1, At, Ak);
                T::mid(Gp, Gt, Gk);
                T::mid(gp, gt, gk);
                T::axpy(X, -h, k1, tX);
                T::axpy(x, -h, f1, tx);
                T::bwd_rhs(Ak, tX, Gk, k2);
                T::bwd_rhs_vec(Ak, tx, gk, f2);
                T::axpy(X, -dt, k2, Xn);
                T::axpy(x, -dt, f2, xn);
            } else {
                double Ak[DD], Gk[DD], gk[D], tX[DD], tx[D], L1[DD], L2[DD], L3[DD], L4[DD], K1[D], K2[D], K3[D], K4[D];
                T::mid(Ap1, At, Ak);
                T::mid(Gp, Gt, Gk);
                T::mid(gp, gt, gk);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L1[q] = k1[q] * dt; tX[q] = X[q] - 0.5 * L1[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K1[q] = f1[q] * dt; tx[q] = x[q] - 0.5 * K1[q]; }
                T::bwd_rhs(Ak, tX, Gk, L2);
                T::bwd_rhs_vec(Ak, tx, gk, K2);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L2[q] *= dt; tX[q] = X[q] - 0.5 * L2[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K2[q] *= dt; tx[q] = x[q] - 0.5 * K2[q]; }
                T::bwd_rhs(Ak, tX, Gk, L3);
                T::bwd_rhs_vec(Ak, tx, gk, K3);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L3[q] *= dt; tX[q] = X[q] - L3[q]; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K3[q] *= dt; tx[q] = x[q] - K3[q]; }
                T::bwd_rhs(Ap1, tX, Gp, L4);
                T::bwd_rhs_vec(Ap1, tx, gp, K4);
#pragma unroll
                for (int q = 0; q < DD; ++q) { L4[q] *= dt; Xn[q] = X[q] - (L1[q] + 2.0 * (L2[q] + L3[q]) + L4[q]) / 6.0; }
#pragma unroll
                for (int q = 0; q < D; ++q) { K4[q] *= dt; xn[q] = x[q] - (K1[q] + 2.0 * (K2[q] + K3[q]) + K4[q]) / 6.0; }
            }
#pragma unroll
            for (int q = 0; q < DD; ++q) X[q] = Xn[q] + jd[q];
#pragma unroll
            for (int q = 0; q < D; ++q) x[q] = xn[q] + jv[q];
            T::store(Po + (size_t)(t - 1) * DD, X);
            T::store(lo + (size_t)(t - 1) * D, x);
            // slot u is free: fetch row t - RING (clamped)
            const int r = t - RING > 0 ? t - RING : 0;
            T::load(Ap + (size_t)r * DD, Ar[u]);
            T::load(gS + (size_t)r * DD, Gr[u]);
            T::load(gm + (size_t)r * D, gr[u]);
            T::load(jmp + (size_t)r * D, jr[u]);
            fr[u] = flagged ? obs_flag[r] : 0;
        }
    }
}
#endif  // VGPA_SMALL_D_SCANS

#ifdef VGPA_SMALL_D_GRAD
// ---------------------------------------------------------------------------
// gradient assembly, one lane per (problem, step): core/gradL_Ab.py:156-177.
// Same sums in the same order as the workgroup kernel.
// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
grad_assemble_small_kernel(size_t total, int N, const double* __restrict__ A, const double* __restrict__ b,
                           const double* __restrict__ m, const double* __restrict__ S,
                           const double* __restrict__ lam, const double* __restrict__ Psi,
                           const double* __restrict__ Ef, const double* __restrict__ Edf, int edf_system,
                           const double* __restrict__ theta, int theta_stride,
                           const double* __restrict__ SigInv, int sig_stride, int sig_diag, double dt, int flags,
                           double* __restrict__ gLA, double* __restrict__ gLb) {
    using T = SmallD<D>;
    constexpr int DD = T::DD;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e / N);
    double Av[DD], Sv[DD], Pv[DD], SI[DD], mv[D], lv[D], uv[D], dbv[D];
    T::load(A + e * DD, Av);
    T::load(S + e * DD, Sv);
    T::load(Psi + e * DD, Pv);
    T::load(SigInv + (size_t)p * sig_stride, SI);
    T::load(m + e * D, mv);
    T::load(lam + e * D, lv);
    const double* th = theta ? theta + (size_t)p * theta_stride : nullptr;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double am = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) am = fma(Av[i * D + k], mv[k], am);
        uv[i] = -Ef[e * D + i] - am + b[e * D + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double d;
        if (sig_diag) {
            d = uv[i] * SI[i * D + i];
        } else {
            d = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) d = fma(uv[k], SI[i * D + k], d);
        }
        dbv[i] = d;
        gLb[e * D + i] = dt * (d + lv[i]);
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        s1 = fma(dbv[i], mv[i], s1);
        s2 = fma(lv[i], mv[i], s2);
    }
    double Y[DD], X[DD];
#pragma unroll
    for (int q = 0; q < DD; ++q)
        Y[q] = (Edf ? Edf[e * DD + q] : edf_implicit(edf_system, D, q / D, q % D, mv, th)) + Av[q];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double s;
            if (sig_diag) {
                s = SI[i * D + i] * Y[i * D + j];
            } else {
                s = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) s = fma(SI[i * D + k], Y[k * D + j], s);
            }
            X[i * D + j] = s - 2.0 * Pv[i * D + j];
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fma(X[i * D + k], Sv[k * D + j], acc);
            const double corr = (flags & VGPA_GRAD_OUTER) ? dbv[i] * mv[j] + lv[i] * mv[j] : s1 + s2;
            gLA[e * DD + i * D + j] = dt * (acc - corr);
        }
}
#endif  // VGPA_SMALL_D_GRAD
