// Heun, RK2 and RK4 scans at D = 40 with symmetric state on the FP64 matrix cores.
//
// Same division of labour as the Euler kernels of scan_mfma.h -- three matrix wavefronts own the
// tile rows of the one product of a right-hand side (A X forward, X A backward; the other half is
// its transpose), a fourth wavefront moves data and integrates the vector state -- with the
// right-hand side evaluated once per stage: two barriers per stage, 2 or 4 stages per step.
//
//   stage inputs      Heun            RK2                 RK4
//   matrix A          A[t], A[t']     A[t], Amid          A[t], Amid, Amid, A[t']
//   (t' = t + 1 forward, t - 1 backward; Amid = 0.5 (A[t] + A[t']), formed by the fourth wavefront
//   from the registers it stages A through, so LDS holds A[t], A[t'] and Amid only)
//
// Forward (core/fwd_ode.py: heun_nD :181, rk2_nD :311, rk4_nD :426): the LDS image of S is
// overwritten by every stage's argument; S(t) is written to HBM out of that image during the
// first stage of step t.  Backward (core/bwd_ode.py: heun_nD :177, rk2_nD :305, rk4_nD :427):
// the matrix wavefronts read their own entries of dEsde_dS(t) from HBM two rows ahead (rows t
// and t - 1 and their mean are the stage inputs), everything else goes through the fourth
// wavefront.
//
// Included by scan.hip after scan_mfma.h, inside its anonymous namespace.

template <int METHOD>
struct M40S {
    static constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD;
    static constexpr bool MID = (METHOD == VGPA_RK2 || METHOD == VGPA_RK4);
    static constexpr int NS = (METHOD == VGPA_RK4) ? 4 : 2;
    // two A buffers, the state image, the transpose buffer, Amid where the method has one
    static constexpr int OFF_X = 2 * MS;
    static constexpr int OFF_MT = 3 * MS;
    static constexpr int OFF_MID = 4 * MS;
    static constexpr int OFF_VEC = (4 + (MID ? 1 : 0)) * MS;
    static constexpr int LDS_DOUBLES = OFF_VEC + 2 * D;
    // LDS offset of the matrix stage `st` multiplies with; `cur` is the buffer (0 / 1) of A[t]
    static __device__ __forceinline__ int stage_a(int st, int cur) {
        if (st == 0) return cur * MS;
        if (METHOD == VGPA_HEUN) return (cur ^ 1) * MS;
        if (METHOD == VGPA_RK2) return OFF_MID;
        return st == 3 ? (cur ^ 1) * MS : OFF_MID;
    }
    // which of (row t, mean, row t') a stage's vector / matrix input is: 0, 1, 2
    static __device__ __forceinline__ int stage_in(int st) {
        if (st == 0) return 0;
        if (METHOD == VGPA_HEUN) return 2;
        if (METHOD == VGPA_RK2) return 1;
        return st == 3 ? 2 : 1;
    }
};

// One stage of the integrators on one scalar entry.  `k` is the right-hand side of stage ST
// evaluated at the previous stage's argument, `x0` the state at the start of the step, `sgn`
// +1 forward and -1 backward.  Returns the argument of the next stage, or the new state after
// the last one.  r1 / r2 carry the stage results a later stage needs.
//
// No contraction into fused multiply-adds here or where the right-hand sides are formed: the
// compiler decides that per unrolled copy, and entries (i, j) and (j, i) of the state live in
// different copies -- with mixed decisions the state is symmetric to rounding only, not bit for bit.
template <int METHOD, int ST>
__device__ __forceinline__ double m40s_stage(double x0, double k, double dt, double sgn, double& r1, double& r2) {
#pragma clang fp contract(off)
    const double h = 0.5 * dt;
    if (METHOD == VGPA_HEUN) {
        if (ST == 0) { r1 = k; return x0 + sgn * (k * dt); }
        return x0 + sgn * (h * (r1 + k));
    } else if (METHOD == VGPA_RK2) {
        if (ST == 0) return x0 + sgn * (h * k);
        return x0 + sgn * (k * dt);
    } else {
        const double Lk = k * dt;
        if (ST == 0) { r1 = Lk; return x0 + sgn * (0.5 * Lk); }
        if (ST == 1) { r2 = Lk; return x0 + sgn * (0.5 * Lk); }
        if (ST == 2) { r2 = r2 + Lk; return x0 + sgn * Lk; }
        return x0 + sgn * ((r1 + 2.0 * r2 + Lk) / 6.0);
    }
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(M40::NT, 2)
fwd_stages_sym_mfma40_kernel(int N, const double* __restrict__ A, const double* __restrict__ b,
                             const double* __restrict__ m0, const double* __restrict__ S0,
                             const double* __restrict__ Sigma, int sigma_stride, double dt,
                             double* __restrict__ m_out, double* __restrict__ S_out) {
    using C = M40S<METHOD>;
    constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD, NS = C::NS;
    extern __shared__ double lds[];
    double* Sb = lds + C::OFF_X;      // S(t), then the argument of each stage
    double* Mt = lds + C::OFF_MT;     // A S, read back transposed
    double* mvv = lds + C::OFF_VEC;   // m(t), then the argument of each stage
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    const double* bp = b + (size_t)p * N * D;
    const double* S0p = S0 + (size_t)p * DD;
    const double* Sgp = Sigma + (size_t)p * sigma_stride;
    double* mo = m_out + (size_t)p * N * D;
    double* So = S_out + (size_t)p * N * DD;
    const int vl = min(lane, D - 1);
    const int r1 = N > 1 ? 1 : 0;

    for (int e = tid; e < DD; e += M40::NT) {
        const int o = (e / D) * LD + (e % D);
        const double a0 = Ap[e], a1 = Ap[(size_t)r1 * DD + e];
        lds[o] = a0;
        lds[MS + o] = a1;
        if (C::MID) lds[C::OFF_MID + o] = 0.5 * (a0 + a1);
        const double sv = S0p[e];
        Sb[o] = sv;
        So[e] = sv;
    }

    if (wid < 3) {
        // ---- matrix wavefronts ----------------------------------------------------------
        const M40Lane L(wid, lane);
        double s[3][4], sig[3][4];
        M40_FOR(c, reg) {
            s[c][reg] = S0p[L.g[c][reg]];
            sig[c][reg] = Sgp[L.g[c][reg]];
        }
        int cur = 0;
#pragma unroll 1
        for (int t = 0; t + 1 < N; ++t) {
            double ra[3][4], rb[3][4];
            auto stage = [&](auto st_tag) __attribute__((always_inline)) {
                constexpr int ST = decltype(st_tag)::value;
                __syncthreads();               // the stage's A and argument are in LDS
                mfma_f64x4 acc[3];
                m40_rows_x_rows(lds + C::stage_a(ST, cur), Sb, wid, lane, acc);
                M40_FOR(c, reg) Mt[L.l[c][reg]] = acc[c][reg];
                __syncthreads();               // A X complete; the stage's inputs are no longer read
                M40_FOR(c, reg) {
#pragma clang fp contract(off)
                    const double mt = Mt[L.lt[c][reg]];
                    const double k = -acc[c][reg] - mt + sig[c][reg];              // core/fwd_ode.py:45
                    const double x = m40s_stage<METHOD, ST>(s[c][reg], k, dt, 1.0, ra[c][reg], rb[c][reg]);
                    if (ST == NS - 1) s[c][reg] = x;
                    Sb[L.l[c][reg]] = x;
                }
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (NS == 4) {
                stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{});
            }
            cur ^= 1;
        }
        __syncthreads();
    } else {
        // ---- fourth wavefront: stages A, integrates m, writes S(t) out of LDS -------------
        double mreg = m0[(size_t)p * D + vl];
        if (lane < D) {
            mvv[lane] = mreg;
            mo[lane] = mreg;
        }
        // An holds A[t+1] (already in LDS; kept for the mean), pre holds A[t+2]
        double An[2 * M40Wave::NQ], pre[2 * M40Wave::NQ];
        const int r2 = N > 2 ? 2 : N - 1;
        if (C::MID) M40Wave::fetch(Ap + (size_t)r1 * DD, lane, An);
        M40Wave::fetch(Ap + (size_t)r2 * DD, lane, pre);
        double b0 = bp[vl], b1 = bp[(size_t)r1 * D + vl], bpre = bp[(size_t)r2 * D + vl];
        int cur = 0;
#pragma unroll 1
        for (int t = 0; t + 1 < N; ++t) {
            double ra = 0.0, rb = 0.0, marg = mreg;
            auto stage = [&](auto st_tag) __attribute__((always_inline)) {
                constexpr int ST = decltype(st_tag)::value;
                __syncthreads();
                if (ST == 0 && t > 0) M40Wave::write_out(Sb, lane, So + (size_t)t * DD);
                if (lane < D) {
                    const double* row = lds + C::stage_a(ST, cur) + lane * LD;
                    double d0 = 0.0, d1 = 0.0;
#pragma unroll 5
                    for (int k = 0; k < D; k += 2) {
                        const double2 av = *reinterpret_cast<const double2*>(row + k);
                        d0 = fma(av.x, mvv[k], d0);
                        d1 = fma(av.y, mvv[k + 1], d1);
                    }
                    const int in = C::stage_in(ST);
                    const double bs = in == 0 ? b0 : (in == 2 ? b1 : 0.5 * (b0 + b1));
                    const double f = -(d0 + d1) + bs;                              // core/fwd_ode.py:44
                    marg = m40s_stage<METHOD, ST>(mreg, f, dt, 1.0, ra, rb);
                }
                __syncthreads();
                if (lane < D) mvv[lane] = marg;
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (NS == 4) {
                stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{});
            }
            // end of the step: no A buffer is being read.  A[t+2] replaces A[t], the mean moves on.
            mreg = marg;
            if (lane < D) mo[(size_t)(t + 1) * D + lane] = mreg;
            M40Wave::commit<false>(pre, lane, lds + cur * MS);
            if (C::MID) {
#pragma unroll
                for (int q = 0; q < 2 * M40Wave::NQ; ++q) {
                    const double mid = 0.5 * (An[q] + pre[q]);
                    An[q] = pre[q];
                    pre[q] = mid;
                }
                M40Wave::commit<false>(pre, lane, lds + C::OFF_MID);
            }
            b0 = b1;
            b1 = bpre;
            const int tp = (t + 3 < N) ? t + 3 : N - 1;
            M40Wave::fetch(Ap + (size_t)tp * DD, lane, pre);
            bpre = bp[(size_t)tp * D + vl];
            cur ^= 1;
        }
        __syncthreads();
        if (N > 1) M40Wave::write_out(Sb, lane, So + (size_t)(N - 1) * DD);
    }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(M40::NT, 2)
bwd_stages_sym_mfma40_kernel(int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                             const double* __restrict__ dEdS, const double* __restrict__ jm,
                             const int* __restrict__ obs_flag, const double* __restrict__ jumpS,
                             int jumpS_stride, double dt, const double* __restrict__ lam_last,
                             const double* __restrict__ Psi_last, double* __restrict__ lam_out,
                             double* __restrict__ Psi_out) {
    using C = M40S<METHOD>;
    constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD, NS = C::NS;
    extern __shared__ double lds[];
    double* Pb = lds + C::OFF_X;      // Psi(t), then the argument of each stage
    double* Mt = lds + C::OFF_MT;     // Psi A, read back transposed
    double* lamv = lds + C::OFF_VEC;  // lam(t), then the argument of each stage
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * DD;
    const double* jmp = jm + (size_t)p * N * D;
    double* lo = lam_out + (size_t)p * N * D;
    double* Po = Psi_out + (size_t)p * N * DD;
    const int vl = min(lane, D - 1);
    const bool jumps = jumpS && obs_flag;
    const double* jSp = jumps ? jumpS + (size_t)p * jumpS_stride : nullptr;
    const int r1 = N > 1 ? N - 2 : 0, r2 = N > 2 ? N - 3 : 0;

    {
        const double* A0 = Ap + (size_t)(N - 1) * DD;
        const double* A1 = Ap + (size_t)r1 * DD;
        double* P0 = Po + (size_t)(N - 1) * DD;
        for (int e = tid; e < DD; e += M40::NT) {
            const int row = e / D, col = e % D;
            const double a0 = A0[e], a1 = A1[e];
            lds[col * LD + row] = a0;                     // the A buffers hold A transposed
            lds[MS + col * LD + row] = a1;
            if (C::MID) lds[C::OFF_MID + col * LD + row] = 0.5 * (a0 + a1);
            const double pv = Psi_last ? Psi_last[(size_t)p * DD + e] : 0.0;
            Pb[row * LD + col] = pv;
            P0[e] = pv;
        }
    }

    if (wid < 3) {
        // ---- matrix wavefronts: their entries of dEsde_dS come straight from HBM, two rows ahead ----
        const M40Lane L(wid, lane);
        double s[3][4], Gt[3][4], Gp[3][4], Gn[3][4];
        M40_FOR(c, reg) {
            s[c][reg] = Psi_last ? Psi_last[(size_t)p * DD + L.g[c][reg]] : 0.0;
            Gt[c][reg] = gS[(size_t)(N - 1) * DD + L.g[c][reg]];
            Gp[c][reg] = gS[(size_t)r1 * DD + L.g[c][reg]];
            Gn[c][reg] = gS[(size_t)r2 * DD + L.g[c][reg]];
        }
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 1; --t) {
            const bool on = jumps && obs_flag[t - 1];
            double ra[3][4], rb[3][4];
            auto stage = [&](auto st_tag) __attribute__((always_inline)) {
                constexpr int ST = decltype(st_tag)::value;
                __syncthreads();
                mfma_f64x4 acc[3];
                m40_rows_x_rows(Pb, lds + C::stage_a(ST, cur), wid, lane, acc);
                M40_FOR(c, reg) Mt[L.l[c][reg]] = acc[c][reg];
                __syncthreads();
                M40_FOR(c, reg) {
#pragma clang fp contract(off)
                    const double mt = Mt[L.lt[c][reg]];
                    const int in = C::stage_in(ST);
                    const double G = in == 0 ? Gt[c][reg] : (in == 2 ? Gp[c][reg] : 0.5 * (Gp[c][reg] + Gt[c][reg]));
                    const double k = -G + (acc[c][reg] + mt);                      // core/bwd_ode.py:43
                    double x = m40s_stage<METHOD, ST>(s[c][reg], k, dt, -1.0, ra[c][reg], rb[c][reg]);
                    if (ST == NS - 1) {
                        if (on) x += jSp[L.g[c][reg]];     // observed rows only; uniform branch
                        s[c][reg] = x;
                    }
                    Pb[L.l[c][reg]] = x;
                }
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (NS == 4) {
                stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{});
            }
            const int tp = (t >= 3) ? t - 3 : 0;
            M40_FOR(c, reg) {
                Gt[c][reg] = Gp[c][reg];
                Gp[c][reg] = Gn[c][reg];
                Gn[c][reg] = gS[(size_t)tp * DD + L.g[c][reg]];
            }
            cur ^= 1;
        }
        __syncthreads();
    } else {
        // ---- fourth wavefront: stages A', integrates lam, writes Psi(t) out of LDS ----------
        double lv = lam_last ? lam_last[(size_t)p * D + vl] : 0.0;
        if (lane < D) {
            lamv[lane] = lv;
            lo[(size_t)(N - 1) * D + lane] = lv;
        }
        double An[2 * M40Wave::NQ], pre[2 * M40Wave::NQ];
        if (C::MID) M40Wave::fetch(Ap + (size_t)r1 * DD, lane, An);
        M40Wave::fetch(Ap + (size_t)r2 * DD, lane, pre);
        double g0 = gm[(size_t)(N - 1) * D + vl], g1 = gm[(size_t)r1 * D + vl], gpre = gm[(size_t)r2 * D + vl];
        double jcur = jmp[(size_t)r1 * D + vl], jpre = jmp[(size_t)r2 * D + vl];
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 1; --t) {
            double ra = 0.0, rb = 0.0, larg = lv;
            auto stage = [&](auto st_tag) __attribute__((always_inline)) {
                constexpr int ST = decltype(st_tag)::value;
                __syncthreads();
                if (ST == 0 && t < N - 1) M40Wave::write_out(Pb, lane, Po + (size_t)t * DD);
                if (lane < D) {
                    const double* At = lds + C::stage_a(ST, cur);
                    double d0 = 0.0, d1 = 0.0;     // (A lam)[lane] = sum_q At[q][lane] lam[q]
#pragma unroll 5
                    for (int k = 0; k < D; k += 2) {
                        d0 = fma(At[k * LD + lane], lamv[k], d0);
                        d1 = fma(At[(k + 1) * LD + lane], lamv[k + 1], d1);
                    }
                    const int in = C::stage_in(ST);
                    const double gs = in == 0 ? g0 : (in == 2 ? g1 : 0.5 * (g1 + g0));
                    const double f = -gs + (d0 + d1);                              // core/bwd_ode.py:42
                    larg = m40s_stage<METHOD, ST>(lv, f, dt, -1.0, ra, rb);
                    if (ST == NS - 1) larg += jcur;                                // core/bwd_ode.py:121-124
                }
                __syncthreads();
                if (lane < D) lamv[lane] = larg;
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (NS == 4) {
                stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{});
            }
            lv = larg;
            if (lane < D) lo[(size_t)(t - 1) * D + lane] = lv;
            M40Wave::commit<true>(pre, lane, lds + cur * MS);
            if (C::MID) {
#pragma unroll
                for (int q = 0; q < 2 * M40Wave::NQ; ++q) {
                    const double mid = 0.5 * (An[q] + pre[q]);
                    An[q] = pre[q];
                    pre[q] = mid;
                }
                M40Wave::commit<true>(pre, lane, lds + C::OFF_MID);
            }
            g0 = g1;
            g1 = gpre;
            jcur = jpre;
            const int tp = (t >= 3) ? t - 3 : 0;
            M40Wave::fetch(Ap + (size_t)tp * DD, lane, pre);
            gpre = gm[(size_t)tp * D + vl];
            jpre = jmp[(size_t)tp * D + vl];
            cur ^= 1;
        }
        __syncthreads();
        if (N > 1) M40Wave::write_out(Pb, lane, Po);
    }
}
