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
    // One step of the forward moment equations, row t -> t + 1 (At, bt: row t; An, bn: row t + 1).
    // core/fwd_ode.py euler_nD :121-131, heun_nD :218-243, rk2_nD :349-363, rk4_nD :460-489.
    template <int METHOD>
    static __device__ __forceinline__ void fwd_step(const double (&At)[DD], const double (&bt)[D], const double (&An)[DD],
                                                    const double (&bn)[D], const double (&Sg)[DD], double dt,
                                                    double (&S)[DD], double (&m)[D]) {
        using T = SmallD<D>;
        const double h = 0.5 * dt;
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
    }
    // One step of the backward equations without the jumps, row t -> t - 1 (At, Gt, gt: row t;
    // Ap1, Gp, gp: row t - 1).  core/bwd_ode.py euler_nD :117-127, heun_nD :211-237, rk2_nD :338-362,
    // rk4_nD :458-493.
    template <int METHOD>
    static __device__ __forceinline__ void bwd_step(const double (&At)[DD], const double (&Gt)[DD], const double (&gt)[D],
                                                    const double (&Ap1)[DD], const double (&Gp)[DD], const double (&gp)[D],
                                                    double dt, double (&X)[DD], double (&x)[D]) {
        using T = SmallD<D>;
        const double h = 0.5 * dt;
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
        for (int q = 0; q < DD; ++q) X[q] = Xn[q];
#pragma unroll
        for (int q = 0; q < D; ++q) x[q] = xn[q];
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
            T::template fwd_step<METHOD>(At, bt, An, bn, Sg, dt, S, m);
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
            T::template bwd_step<METHOD>(At, Gt, gt, Ap1, Gp, gp, dt, X, x);
#pragma unroll
            for (int q = 0; q < DD; ++q) X[q] += jd[q];
#pragma unroll
            for (int q = 0; q < D; ++q) x[q] += jv[q];
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

// ---------------------------------------------------------------------------
// The same scans with their global traffic staged through LDS.
//
// In the kernels above every lane walks the rows of its own problem, so each load or store
// instruction touches 64 different cache lines and the address path, not the arithmetic, sets
// the time.  Here one wavefront owns PB = 16 problems and moves their rows in chunks of T = 8
// time steps: four lanes copy the (contiguous) chunk of each problem with 16-byte accesses,
// LDS holds the chunk problem by problem with an odd stride, and lane p < PB integrates
// problem p out of LDS.  The inputs of the next chunk are in flight (in registers) while the
// current one is integrated; results go back through LDS the same way.  Sixteen problems per
// wavefront also spread a grid of a few thousand problems over every compute unit.
//
// Pieces are 16 bytes, so N * D must be even (N * D * D then is too); other sizes take the
// kernels above.  Same step arithmetic, same bits.
// ---------------------------------------------------------------------------
struct SmallLds {
    static constexpr int PB = 16, T = 8, R = 10;     // problems per wavefront, steps and staged rows per chunk
    // Four lanes move the chunk of one problem: lane l belongs to problem l / 4 of the wavefront's sixteen and
    // moves the 16-byte pieces 4 k + l % 4, k = 0, 1, ... of that problem's chunk.  A lane's global and LDS
    // addresses are then one base each plus a compile-time offset per instruction; an instruction touches a
    // 64-byte segment per problem, and consecutive instructions complete the cache lines.
    static __device__ __forceinline__ int prob_of(int lane) { return lane >> 2; }
    static __device__ __forceinline__ int sub_of(int lane) { return lane & 3; }
    // Rows [row0, row0 + R) of a (P, N, RD) array.  `pbase` is the element offset of this lane's problem
    // (clamped to the last problem), EDGE chunks clamp rows outside [0, N) to the edge rows.
    template <int RD>
    struct In {
        static constexpr int CH = R * RD, CH2 = CH / 2, STRIDE = CH | 1, NI = (CH2 + 3) / 4;
        static constexpr int LDS = PB * STRIDE;
        template <bool EDGE>
        static __device__ __forceinline__ void fetch(const double* __restrict__ src, size_t pbase, long long per,
                                                     long long row0, int lane, double (&pre)[2 * NI]) {
            const int sub = sub_of(lane);
            const double* base = src + pbase + row0 * RD + 2 * sub;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int e2 = (4 * k + 3 < CH2) ? 4 * k + sub : min(4 * k + sub, CH2 - 1);
                double2 v;
                if (EDGE) {
                    long long off = row0 * RD + 2 * e2;
                    off = off < 0 ? 0 : (off > per - 2 ? per - 2 : off);      // rows outside [0, N) repeat the edge
                    v = *reinterpret_cast<const double2*>(src + pbase + off);
                } else {
                    v = *reinterpret_cast<const double2*>(base + 2 * (e2 - sub));
                }
                pre[2 * k] = v.x;
                pre[2 * k + 1] = v.y;
            }
        }
        static __device__ __forceinline__ void commit(const double (&pre)[2 * NI], int lane, double* dst) {
            const int sub = sub_of(lane);
            double* base = dst + prob_of(lane) * STRIDE + 2 * sub;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int e2 = (4 * k + 3 < CH2) ? 4 * k + sub : min(4 * k + sub, CH2 - 1);   // tail lanes repeat a piece
                base[2 * (e2 - sub)] = pre[2 * k];
                base[2 * (e2 - sub) + 1] = pre[2 * k + 1];
            }
        }
    };
    // Rows [row0, row0 + T) of a (P, N, RD) array, LDS -> global.  `valid`: this lane's problem exists;
    // EDGE chunks skip rows >= N.
    template <int RD>
    struct Out {
        static constexpr int CH = T * RD, CH2 = CH / 2, STRIDE = CH | 1, NI = CH2 / 4;
        static_assert(CH2 % 4 == 0, "whole groups of four pieces");
        static constexpr int LDS = PB * STRIDE;
        template <bool EDGE>
        static __device__ __forceinline__ void flush(const double* src, double* __restrict__ dst, size_t pbase, long long per,
                                                     long long row0, bool valid, int lane) {
            const int sub = sub_of(lane);
            const double* sb = src + prob_of(lane) * STRIDE + 2 * sub;
            double* db = dst + pbase + row0 * RD + 2 * sub;
            if (valid) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    if (!EDGE || row0 * RD + 2 * (4 * k + sub) + 1 < per)
                        *reinterpret_cast<double2*>(db + 8 * k) = make_double2(sb[8 * k], sb[8 * k + 1]);
                }
            }
        }
    };
    static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
};

template <int D, int METHOD>
__global__ void __launch_bounds__(64, 1)
fwd_scan_small_lds_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ b,
                          const double* __restrict__ m0, const double* __restrict__ S0,
                          const double* __restrict__ Sigma, int sigma_stride, double dt,
                          double* __restrict__ m_out, double* __restrict__ S_out) {
    using Tm = SmallD<D>;
    using L = SmallLds;
    constexpr int DD = Tm::DD, PB = L::PB, T = L::T;
    using InA = L::In<DD>;
    using Inb = L::In<D>;
    using OutS = L::Out<DD>;
    using Outm = L::Out<D>;
    __shared__ double lds[InA::LDS + Inb::LDS + OutS::LDS + Outm::LDS];
    double* inA = lds;
    double* inb = inA + InA::LDS;
    double* outS = inb + Inb::LDS;
    double* outm = outS + OutS::LDS;
    const int lane = threadIdx.x;
    const int p0 = blockIdx.x * PB;
    const bool act = lane < PB && p0 + lane < P;
    const int p = min(p0 + min(lane, PB - 1), P - 1);
    double Sg[DD], S[DD], m[D];
    Tm::load(Sigma + (size_t)p * sigma_stride, Sg);
    Tm::load(S0 + (size_t)p * DD, S);
    Tm::load(m0 + (size_t)p * D, m);
    // staging: this lane's problem (four lanes per problem) and its element offsets in the arrays
    const int sp = min(p0 + L::prob_of(lane), P - 1);
    const bool svalid = p0 + L::prob_of(lane) < P;
    const long long perA = (long long)N * DD, perb = (long long)N * D;
    const size_t baseA = (size_t)sp * perA, baseb = (size_t)sp * perb;
    double preA[2 * InA::NI], preb[2 * Inb::NI];
    InA::template fetch<true>(A, baseA, perA, 0, lane, preA);
    Inb::template fetch<true>(b, baseb, perb, 0, lane, preb);
    const double* myA = inA + min(lane, PB - 1) * InA::STRIDE;
    const double* myb = inb + min(lane, PB - 1) * Inb::STRIDE;
    double* myS = outS + min(lane, PB - 1) * OutS::STRIDE;
    double* mym = outm + min(lane, PB - 1) * Outm::STRIDE;
    for (int t0 = 0; t0 < N; t0 += T) {
        InA::commit(preA, lane, inA);
        Inb::commit(preb, lane, inb);
        L::fence();
        // the next chunk, in flight during this one (edge rows are clamped in the last two chunks only)
        if (t0 + T + L::R <= N) {
            InA::template fetch<false>(A, baseA, perA, t0 + T, lane, preA);
            Inb::template fetch<false>(b, baseb, perb, t0 + T, lane, preb);
        } else {
            InA::template fetch<true>(A, baseA, perA, t0 + T, lane, preA);
            Inb::template fetch<true>(b, baseb, perb, t0 + T, lane, preb);
        }
        if (act) {
#pragma unroll
            for (int u = 0; u < T; ++u) {
                // row t0 + u is the current state; then one step with rows u, u + 1 of the chunk
                Tm::store(myS + u * DD, S);
                Tm::store(mym + u * D, m);
                double At[DD], An[DD], bt[D], bn[D];
                Tm::load(myA + u * DD, At);
                Tm::load(myb + u * D, bt);
                if (METHOD != VGPA_EULER) {
                    Tm::load(myA + (u + 1) * DD, An);
                    Tm::load(myb + (u + 1) * D, bn);
                }
                Tm::template fwd_step<METHOD>(At, bt, An, bn, Sg, dt, S, m);
            }
        }
        L::fence();
        if (t0 + T <= N) {
            OutS::template flush<false>(outS, S_out, baseA, perA, t0, svalid, lane);
            Outm::template flush<false>(outm, m_out, baseb, perb, t0, svalid, lane);
        } else {
            OutS::template flush<true>(outS, S_out, baseA, perA, t0, svalid, lane);
            Outm::template flush<true>(outm, m_out, baseb, perb, t0, svalid, lane);
        }
        L::fence();
    }
}

template <int D, int METHOD>
__global__ void __launch_bounds__(64, 1)
bwd_scan_small_lds_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                          const double* __restrict__ dEdS, const double* __restrict__ jm,
                          const int* __restrict__ obs_flag, const double* __restrict__ jumpS, int jumpS_stride,
                          double dt, const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                          double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    using Tm = SmallD<D>;
    using L = SmallLds;
    constexpr int DD = Tm::DD, PB = L::PB, T = L::T;
    using InM = L::In<DD>;
    using Inv = L::In<D>;
    using OutM = L::Out<DD>;
    using Outv = L::Out<D>;
    __shared__ double lds[2 * InM::LDS + 2 * Inv::LDS + OutM::LDS + Outv::LDS];
    double* inA = lds;
    double* inG = inA + InM::LDS;
    double* ing = inG + InM::LDS;
    double* inj = ing + Inv::LDS;
    double* outP = inj + Inv::LDS;
    double* outl = outP + OutM::LDS;
    const int lane = threadIdx.x;
    const int p0 = blockIdx.x * PB;
    const bool act = lane < PB && p0 + lane < P;
    const int pl = min(lane, PB - 1);
    const int p = min(p0 + pl, P - 1);
    const bool flagged = jumpS && obs_flag;
    double jt[DD], X[DD], x[D];
#pragma unroll
    for (int q = 0; q < DD; ++q) {
        jt[q] = flagged ? jumpS[(size_t)p * jumpS_stride + q] : 0.0;
        X[q] = Psi_last ? Psi_last[(size_t)p * DD + q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < D; ++q) x[q] = lam_last ? lam_last[(size_t)p * D + q] : 0.0;
    // chunk c holds rows [T c, T c + T); its inputs are rows [T c - 2, T c + T): row t - 1 is needed with row t
    const int nch = (N + T - 1) / T;
    const int sp = min(p0 + L::prob_of(lane), P - 1);
    const bool svalid = p0 + L::prob_of(lane) < P;
    const long long perM = (long long)N * DD, perv = (long long)N * D;
    const size_t baseM = (size_t)sp * perM, basev = (size_t)sp * perv;
    double preA[2 * InM::NI], preG[2 * InM::NI], preg[2 * Inv::NI], prej[2 * Inv::NI];
    auto fetch = [&](int c) {
        const long long r0 = (long long)T * c - 2;
        if (r0 >= 0 && r0 + L::R <= N) {
            InM::template fetch<false>(A, baseM, perM, r0, lane, preA);
            InM::template fetch<false>(dEdS, baseM, perM, r0, lane, preG);
            Inv::template fetch<false>(dEdm, basev, perv, r0, lane, preg);
            Inv::template fetch<false>(jm, basev, perv, r0, lane, prej);
        } else {
            InM::template fetch<true>(A, baseM, perM, r0, lane, preA);
            InM::template fetch<true>(dEdS, baseM, perM, r0, lane, preG);
            Inv::template fetch<true>(dEdm, basev, perv, r0, lane, preg);
            Inv::template fetch<true>(jm, basev, perv, r0, lane, prej);
        }
    };
    fetch(nch - 1);
    const double* myA = inA + pl * InM::STRIDE;
    const double* myG = inG + pl * InM::STRIDE;
    const double* myg = ing + pl * Inv::STRIDE;
    const double* myj = inj + pl * Inv::STRIDE;
    double* myP = outP + pl * OutM::STRIDE;
    double* myl = outl + pl * Outv::STRIDE;
    for (int c = nch - 1; c >= 0; --c) {
        InM::commit(preA, lane, inA);
        InM::commit(preG, lane, inG);
        Inv::commit(preg, lane, ing);
        Inv::commit(prej, lane, inj);
        L::fence();
        fetch(c > 0 ? c - 1 : 0);
        const int t0 = T * c;
        if (act) {
#pragma unroll
            for (int u = T - 1; u >= 0; --u) {
                const int t = t0 + u;
                if (t < N) {                              // uniform: only the top chunk has rows beyond N - 1
                    Tm::store(myP + u * DD, X);
                    Tm::store(myl + u * D, x);
                    if (t >= 1) {
                        // chunk rows: row t is entry u + 2, row t - 1 entry u + 1
                        double At[DD], Gt[DD], gt[D], Ap1[DD], Gp[DD], gp[D], jv[D];
                        Tm::load(myA + (u + 2) * DD, At);
                        Tm::load(myG + (u + 2) * DD, Gt);
                        Tm::load(myg + (u + 2) * D, gt);
                        if (METHOD != VGPA_EULER) {
                            Tm::load(myA + (u + 1) * DD, Ap1);
                            Tm::load(myG + (u + 1) * DD, Gp);
                            Tm::load(myg + (u + 1) * D, gp);
                        }
                        Tm::load(myj + (u + 1) * D, jv);
                        Tm::template bwd_step<METHOD>(At, Gt, gt, Ap1, Gp, gp, dt, X, x);
                        const bool on = flagged && obs_flag[t - 1] != 0;      // core/bwd_ode.py:124,127
#pragma unroll
                        for (int q = 0; q < DD; ++q) X[q] += on ? jt[q] : 0.0;
#pragma unroll
                        for (int q = 0; q < D; ++q) x[q] += jv[q];
                    }
                }
            }
        }
        L::fence();
        if (t0 + T <= N) {
            OutM::template flush<false>(outP, Psi_out, baseM, perM, t0, svalid, lane);
            Outv::template flush<false>(outl, lam_out, basev, perv, t0, svalid, lane);
        } else {
            OutM::template flush<true>(outP, Psi_out, baseM, perM, t0, svalid, lane);
            Outv::template flush<true>(outl, lam_out, basev, perv, t0, svalid, lane);
        }
        L::fence();
    }
}
// ---------------------------------------------------------------------------
// Euler scans with a ROW of the state per lane.
//
// With a whole problem per lane the time of the kernels above is the dependent FP64 chain of
// one lane (about a hundred instructions per step for D = 3).  Here the four lanes that stage
// a problem's chunk also integrate it: lane i of the quad keeps row i of S (or Psi) and entry
// i of m (or lam), rebuilds the whole state from the rows of its neighbours with DPP quad
// broadcasts at the start of a step, and computes its row of the next state -- a third of the
// arithmetic per lane, all 64 lanes busy.  Same products in the same order as SmallD::fwd_step /
// bwd_step for Euler, so the results are the same bits.  Lanes D .. 3 of a quad repeat row D - 1.
// ---------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = R | (R << 2) | (R << 4) | (R << 6);      // quad_perm:[R,R,R,R]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int D, int R>
struct QuadGather {
    // full[r][:] = row held by lane r of the quad, v[r] = scalar held by lane r
    static __device__ __forceinline__ void run(const double (&row)[D], double s, double (&full)[D * D], double (&v)[D]) {
        if constexpr (R < D) {
#pragma unroll
            for (int j = 0; j < D; ++j) full[R * D + j] = quad_bcast<R>(row[j]);
            v[R] = quad_bcast<R>(s);
            QuadGather<D, R + 1>::run(row, s, full, v);
        }
    }
};

// A workgroup is two wavefronts: the first integrates, the second moves the data -- it keeps the
// inputs of the next DEPTH chunks in flight in its registers, commits a chunk to one of two LDS
// buffers while the first wavefront integrates out of the other, and writes the finished chunk of
// results out of a second pair of buffers.  One barrier per chunk.
struct SmallRows {
    static constexpr int NT = 128;
    static constexpr int DEPTH_FWD = 3, DEPTH_BWD = 2;    // chunks in flight
    static constexpr int in_lds(int rd) { return SmallLds::PB * ((SmallLds::R * rd) | 1); }
    static constexpr int out_lds(int rd) { return SmallLds::PB * ((SmallLds::T * rd) | 1); }
    static constexpr int fwd_lds_doubles(int D) { return 2 * (in_lds(D * D) + in_lds(D)) + 2 * (out_lds(D * D) + out_lds(D)); }
    static constexpr int bwd_lds_doubles(int D) { return 4 * (in_lds(D * D) + in_lds(D)) + 2 * (out_lds(D * D) + out_lds(D)); }
};

template <int D>
__global__ void __launch_bounds__(SmallRows::NT, 1)
fwd_scan_small_rows_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ b,
                           const double* __restrict__ m0, const double* __restrict__ S0,
                           const double* __restrict__ Sigma, int sigma_stride, double dt,
                           double* __restrict__ m_out, double* __restrict__ S_out) {
    using Tm = SmallD<D>;
    using L = SmallLds;
    constexpr int DD = Tm::DD, PB = L::PB, T = L::T, DEPTH = SmallRows::DEPTH_FWD;
    using InA = L::In<DD>;
    using Inb = L::In<D>;
    using OutS = L::Out<DD>;
    using Outm = L::Out<D>;
    constexpr int IN = InA::LDS + Inb::LDS, OUT = OutS::LDS + Outm::LDS;
    extern __shared__ double lds[];                    // 2 IN + 2 OUT doubles: SmallRows::fwd_lds_doubles(D)
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int p0 = blockIdx.x * PB;
    const int pl = L::prob_of(lane);
    const int sp = min(p0 + pl, P - 1);
    const int nch = (N + T - 1) / T;
    if (role == 1) {
        // ---- data movement ----
        const bool svalid = p0 + pl < P;
        const long long perA = (long long)N * DD, perb = (long long)N * D;
        const size_t baseA = (size_t)sp * perA, baseb = (size_t)sp * perb;
        double preA[DEPTH][2 * InA::NI], preb[DEPTH][2 * Inb::NI];
        auto fetch = [&](int c, double (&pa)[2 * InA::NI], double (&pb)[2 * Inb::NI]) {
            const long long r0 = (long long)T * min(c, nch - 1);
            if (r0 + L::R <= N) {
                InA::template fetch<false>(A, baseA, perA, r0, lane, pa);
                Inb::template fetch<false>(b, baseb, perb, r0, lane, pb);
            } else {
                InA::template fetch<true>(A, baseA, perA, r0, lane, pa);
                Inb::template fetch<true>(b, baseb, perb, r0, lane, pb);
            }
        };
        auto flush = [&](int c) {
            const double* out = lds + 2 * IN + (c & 1) * OUT;
            const int t0 = T * c;
            if (t0 + T <= N) {
                OutS::template flush<false>(out, S_out, baseA, perA, t0, svalid, lane);
                Outm::template flush<false>(out + OutS::LDS, m_out, baseb, perb, t0, svalid, lane);
            } else {
                OutS::template flush<true>(out, S_out, baseA, perA, t0, svalid, lane);
                Outm::template flush<true>(out + OutS::LDS, m_out, baseb, perb, t0, svalid, lane);
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, preA[d], preb[d]);
        for (int c0 = 0; c0 < nch; c0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int c = c0 + d;
                if (c < nch) {
                    double* in = lds + (c & 1) * IN;
                    InA::commit(preA[d], lane, in);
                    Inb::commit(preb[d], lane, in + InA::LDS);
                    fetch(c + DEPTH, preA[d], preb[d]);
                    __syncthreads();                  // chunk c is in LDS; the results of chunk c - 1 are complete
                    if (c >= 1) flush(c - 1);
                }
            }
        }
        __syncthreads();
        flush(nch - 1);
        return;
    }
    // ---- integration: lane i of a quad holds row i of S and entry i of m ----
    const int i = min(L::sub_of(lane), D - 1);
    double Sgr[D], Sr[D], mi;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        Sgr[j] = Sigma[(size_t)sp * sigma_stride + i * D + j];
        Sr[j] = S0[(size_t)sp * DD + i * D + j];
    }
    mi = m0[(size_t)sp * D + i];
    for (int c = 0; c < nch; ++c) {
        const double* myA = lds + (c & 1) * IN + pl * InA::STRIDE;
        const double* myb = lds + (c & 1) * IN + InA::LDS + pl * Inb::STRIDE + i;
        double* myS = lds + 2 * IN + (c & 1) * OUT + pl * OutS::STRIDE + i * D;
        double* mym = lds + 2 * IN + (c & 1) * OUT + OutS::LDS + pl * Outm::STRIDE + i;
        __syncthreads();
        // The inputs of step u + 1 are read from LDS before the state of row u is stored there: the
        // compiler cannot move a load above a store that might alias, and a load issued only after the
        // store would put its latency on the chain of every step.
        double At[2][DD], ai[2][D], bi[2];
        auto load_in = [&](int u, int w) {
            Tm::load(myA + u * DD, At[w]);
#pragma unroll
            for (int k = 0; k < D; ++k) ai[w][k] = myA[u * DD + i * D + k];
            bi[w] = myb[u * D];
        };
        load_in(0, 0);
#pragma unroll
        for (int u = 0; u < T; ++u) {
            const int w = u & 1;
            if (u + 1 < T) load_in(u + 1, w ^ 1);
            // row T c + u is the current state; then one step with row u of the chunk
#pragma unroll
            for (int j = 0; j < D; ++j) myS[u * DD + j] = Sr[j];
            mym[u * D] = mi;
            double S[DD], m[D];
            QuadGather<D, 0>::run(Sr, mi, S, m);
            // row i of  S + dt (Sig - A S - S A')  and entry i of  m + dt (-A m + b)   (core/fwd_ode.py:121-131)
            double Sn[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    s1 = fma(ai[w][k], S[k * D + j], s1);
                    s2 = fma(Sr[k], At[w][j * D + k], s2);
                }
                const double K = -s1 - s2 + Sgr[j];
                Sn[j] = Sr[j] + dt * K;
            }
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s = fma(ai[w][k], m[k], s);
            const double f = -s + bi[w];
            mi = mi + dt * f;
#pragma unroll
            for (int j = 0; j < D; ++j) Sr[j] = Sn[j];
        }
    }
    __syncthreads();
}

template <int D>
__global__ void __launch_bounds__(SmallRows::NT, 1)
bwd_scan_small_rows_kernel(int P, int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                           const double* __restrict__ dEdS, const double* __restrict__ jm,
                           const int* __restrict__ obs_flag, const double* __restrict__ jumpS, int jumpS_stride,
                           double dt, const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                           double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    using Tm = SmallD<D>;
    using L = SmallLds;
    constexpr int DD = Tm::DD, PB = L::PB, T = L::T, DEPTH = SmallRows::DEPTH_BWD;
    using InM = L::In<DD>;
    using Inv = L::In<D>;
    using OutM = L::Out<DD>;
    using Outv = L::Out<D>;
    constexpr int IN = 2 * InM::LDS + 2 * Inv::LDS, OUT = OutM::LDS + Outv::LDS;
    extern __shared__ double lds[];                    // 2 IN + 2 OUT doubles: SmallRows::bwd_lds_doubles(D)
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int p0 = blockIdx.x * PB;
    const int pl = L::prob_of(lane);
    const int sp = min(p0 + pl, P - 1);
    // chunk c holds rows [T c, T c + T); its inputs are rows [T c - 2, T c + T): row t - 1 is needed with row t.
    // The chunks are visited from the last to the first: visit v is chunk nch - 1 - v.
    const int nch = (N + T - 1) / T;
    if (role == 1) {
        // ---- data movement ----
        const bool svalid = p0 + pl < P;
        const long long perM = (long long)N * DD, perv = (long long)N * D;
        const size_t baseM = (size_t)sp * perM, basev = (size_t)sp * perv;
        double preA[DEPTH][2 * InM::NI], preG[DEPTH][2 * InM::NI], preg[DEPTH][2 * Inv::NI], prej[DEPTH][2 * Inv::NI];
        auto fetch = [&](int v, double (&pa)[2 * InM::NI], double (&pg)[2 * InM::NI], double (&pv)[2 * Inv::NI],
                         double (&pj)[2 * Inv::NI]) {
            const int c = max(nch - 1 - v, 0);
            const long long r0 = (long long)T * c - 2;
            if (r0 >= 0 && r0 + L::R <= N) {
                InM::template fetch<false>(A, baseM, perM, r0, lane, pa);
                InM::template fetch<false>(dEdS, baseM, perM, r0, lane, pg);
                Inv::template fetch<false>(dEdm, basev, perv, r0, lane, pv);
                Inv::template fetch<false>(jm, basev, perv, r0, lane, pj);
            } else {
                InM::template fetch<true>(A, baseM, perM, r0, lane, pa);
                InM::template fetch<true>(dEdS, baseM, perM, r0, lane, pg);
                Inv::template fetch<true>(dEdm, basev, perv, r0, lane, pv);
                Inv::template fetch<true>(jm, basev, perv, r0, lane, pj);
            }
        };
        auto flush = [&](int v) {
            const double* out = lds + 2 * IN + (v & 1) * OUT;
            const int t0 = T * (nch - 1 - v);
            if (t0 + T <= N) {
                OutM::template flush<false>(out, Psi_out, baseM, perM, t0, svalid, lane);
                Outv::template flush<false>(out + OutM::LDS, lam_out, basev, perv, t0, svalid, lane);
            } else {
                OutM::template flush<true>(out, Psi_out, baseM, perM, t0, svalid, lane);
                Outv::template flush<true>(out + OutM::LDS, lam_out, basev, perv, t0, svalid, lane);
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, preA[d], preG[d], preg[d], prej[d]);
        for (int v0 = 0; v0 < nch; v0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int v = v0 + d;
                if (v < nch) {
                    double* in = lds + (v & 1) * IN;
                    InM::commit(preA[d], lane, in);
                    InM::commit(preG[d], lane, in + InM::LDS);
                    Inv::commit(preg[d], lane, in + 2 * InM::LDS);
                    Inv::commit(prej[d], lane, in + 2 * InM::LDS + Inv::LDS);
                    fetch(v + DEPTH, preA[d], preG[d], preg[d], prej[d]);
                    __syncthreads();
                    if (v >= 1) flush(v - 1);
                }
            }
        }
        __syncthreads();
        flush(nch - 1);
        return;
    }
    // ---- integration: lane i of a quad holds row i of Psi and entry i of lam ----
    const int i = min(L::sub_of(lane), D - 1);
    const bool flagged = jumpS && obs_flag;
    double jtr[D], Xr[D], xi;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        jtr[j] = flagged ? jumpS[(size_t)sp * jumpS_stride + i * D + j] : 0.0;
        Xr[j] = Psi_last ? Psi_last[(size_t)sp * DD + i * D + j] : 0.0;
    }
    xi = lam_last ? lam_last[(size_t)sp * D + i] : 0.0;
    for (int v = 0; v < nch; ++v) {
        const double* in = lds + (v & 1) * IN;
        const double* myA = in + pl * InM::STRIDE;
        const double* myG = in + InM::LDS + pl * InM::STRIDE + i * D;
        const double* myg = in + 2 * InM::LDS + pl * Inv::STRIDE + i;
        const double* myj = in + 2 * InM::LDS + Inv::LDS + pl * Inv::STRIDE + i;
        double* myP = lds + 2 * IN + (v & 1) * OUT + pl * OutM::STRIDE + i * D;
        double* myl = lds + 2 * IN + (v & 1) * OUT + OutM::LDS + pl * Outv::STRIDE + i;
        const int t0 = T * (nch - 1 - v);
        __syncthreads();
        // inputs of the next step are read before the current row is stored (see the forward kernel)
        double At[2][DD], ari[2][D], aci[2][D], Gi[2][D], gi[2], jvi[2];
        bool on[2];
        auto load_in = [&](int u, int w) {
            // chunk rows: row t = t0 + u is entry u + 2, row t - 1 entry u + 1
            Tm::load(myA + (u + 2) * DD, At[w]);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                ari[w][k] = myA[(u + 2) * DD + i * D + k];
                aci[w][k] = myA[(u + 2) * DD + k * D + i];
                Gi[w][k] = myG[(u + 2) * DD + k];
            }
            gi[w] = myg[(u + 2) * D];
            jvi[w] = myj[(u + 1) * D];
            on[w] = flagged && t0 + u >= 1 && t0 + u < N && obs_flag[t0 + u - 1] != 0;      // core/bwd_ode.py:124,127
        };
        load_in(T - 1, (T - 1) & 1);
#pragma unroll
        for (int u = T - 1; u >= 0; --u) {
            const int t = t0 + u;
            const int w = u & 1;
            if (u >= 1) load_in(u - 1, w ^ 1);
            if (t < N) {                              // uniform: only the top chunk has rows beyond N - 1
#pragma unroll
                for (int j = 0; j < D; ++j) myP[u * DD + j] = Xr[j];
                myl[u * D] = xi;
                if (t >= 1) {
                    double X[DD], x[D];
                    QuadGather<D, 0>::run(Xr, xi, X, x);
                    // row i of  Psi - dt (-G + Psi A + A' Psi)  and entry i of  lam - dt (-g + A lam)   (core/bwd_ode.py:117-127)
                    const double ndt = -dt;
                    double Xn[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        double s1 = 0.0, s2 = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            s1 = fma(Xr[k], At[w][k * D + j], s1);
                            s2 = fma(aci[w][k], X[k * D + j], s2);
                        }
                        const double K = -Gi[w][j] + (s1 + s2);
                        Xn[j] = Xr[j] + ndt * K;
                        Xn[j] += on[w] ? jtr[j] : 0.0;
                    }
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) s = fma(ari[w][k], x[k], s);
                    const double f = -gi[w] + s;
                    xi = xi + ndt * f;
                    xi += jvi[w];
#pragma unroll
                    for (int j = 0; j < D; ++j) Xr[j] = Xn[j];
                }
            }
        }
    }
    __syncthreads();
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
