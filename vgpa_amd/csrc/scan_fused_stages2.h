// Backward Heun and RK2 scans at D = 40 with symmetric state, fused with the gradient assembly: the
// two-stage counterparts of bwd_grad_euler_sym_mfma40_kernel (scan_fused.h), same eight roles.
//
// A step t -> t-1 has two stages and four workgroup barriers.  Heun (core/bwd_ode.py:177-238):
//   stage 0   k1 = -G(t) + Psi(t) A(t) + A(t)' Psi(t)          predictor  X = Psi(t) - dt k1
//   stage 1   k2 = -G(t-1) + X A(t-1) + A(t-1)' X              Psi(t-1) = Psi(t) - dt/2 (k1 + k2) + jump
// RK2 (core/bwd_ode.py:305-364), with Am = (A(t) + A(t-1))/2 and Gm = (G(t) + G(t-1))/2:
//   stage 0   k1 as above                                      X = Psi(t) - dt/2 k1
//   stage 1   k2 = -Gm + X Am + Am' X                           Psi(t-1) = Psi(t) - dt k2 + jump
// The gradient of row t (core/gradL_Ab.py:156-177) is formed during stage 0, while A(t)',
// Psi(t), lam(t), S(t), m(t) are the LDS images.  LDS holds five matrices as in the Euler kernel:
// Psi / predictor, A(t)', the matrix of stage 1 (A(t-1)', which is the next step's A(t)': the two
// buffers alternate; for RK2 Am', formed by the fourth wavefront from the A(t-1)' it has just put
// into LDS and the A(t-2) it holds in registers), the transpose buffer, and one buffer that is S(t) during stage
// 0 (read by the gradient wavefronts) and dEsde_dS(t-1) during stage 1 (read by the recurrence
// wavefronts, which keep it in registers as the G(t) of the next step).
//
// The recurrence wavefronts have no registers left for k1 next to Psi(t) (128 per wavefront with two
// workgroups per CU), so for Heun they carry Psi(t) - dt/2 k1 and subtract dt/2 k2 from it: Psi is
// equal to the separate scan's to rounding, not bit for bit, and stays exactly symmetric.
//
// Included by scan.hip after scan_fused.h, inside its anonymous namespace.

struct M40FH {
    static constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD;
    static constexpr int NT = 512;
    static constexpr int OFF_AT = MS;               // A(t)' and the matrix of stage 1
    static constexpr int OFF_MT = 3 * MS;
    static constexpr int OFF_SG = 4 * MS;           // S(t) in stage 0, dEsde_dS(t-1) in stage 1
    static constexpr int OFF_LAM = 5 * MS;
    static constexpr int OFF_MX = OFF_LAM + D;      // m(t) on the ring: mx[k] = m[(k - 2) mod D], k = 0 .. 47
    static constexpr int OFF_SI = OFF_MX + 48;      // diagonal of SigInv
    static constexpr int OFF_SC = OFF_SI + D;       // s1 + s2 of the current row
    static constexpr int LDS_DOUBLES = OFF_SC + 2;
};

// the pieces of a transposed LDS image that M40Rows::commit<true> wrote, read back by the lane that wrote them
__device__ __forceinline__ void m40rows_mid_transposed(const double* prev_img, const double (&pre)[2 * M40Rows::NQ], int lane,
                                                       double* mid_img) {
    int lr, lc;
    M40Rows::split(lane, lr, lc);
    const int o0 = 2 * lc * M40::LD + lr;
#pragma unroll
    for (int i = 0; i < M40Rows::NQ; ++i) {
        const int o = (i < M40Rows::NQ - 1) ? o0 + 3 * i : o0 + 3 * i - lr;
        mid_img[o] = 0.5 * (prev_img[o] + pre[2 * i]);
        mid_img[o + M40::LD] = 0.5 * (prev_img[o + M40::LD] + pre[2 * i + 1]);
    }
}

template <int METHOD>
__global__ void __launch_bounds__(M40FH::NT, 4)
bwd_grad_stages2_sym_mfma40_kernel(int N, const double* __restrict__ A, const double* __restrict__ bvec,
                                const double* __restrict__ mvec, const double* __restrict__ Smat,
                                const double* __restrict__ dEdm, const double* __restrict__ dEdS,
                                const double* __restrict__ jm, const int* __restrict__ obs_flag,
                                const double* __restrict__ jumpS, int jumpS_stride,
                                const double* __restrict__ Efv, const double* __restrict__ SigInv, int sig_stride,
                                double dt, const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                                double* __restrict__ lam_out, double* __restrict__ Psi_out, int psi_all,
                                double* __restrict__ gLA, double* __restrict__ gLb) {
    constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD;
    constexpr bool RK2 = (METHOD == VGPA_RK2);     // otherwise Heun
    extern __shared__ double lds[];
    double* Pb = lds;                       // Psi[t], then the predictor
    double* Mt = lds + M40FH::OFF_MT;       // X A, read back transposed
    double* SG = lds + M40FH::OFF_SG;
    double* lamv = lds + M40FH::OFF_LAM;
    double* mx = lds + M40FH::OFF_MX;
    double* siv = lds + M40FH::OFF_SI;
    double* sc = lds + M40FH::OFF_SC;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    const double* Sp = Smat + (size_t)p * N * DD;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * DD;
    const double* jmp = jm + (size_t)p * N * D;
    const double* bp = bvec + (size_t)p * N * D;
    const double* mp = mvec + (size_t)p * N * D;
    const double* efp = Efv + (size_t)p * N * D;
    double* lo = lam_out + (size_t)p * N * D;
    // psi_all == 0: Psi_out is (P, 1, D, D) and receives row 0 only
    double* Po = Psi_out + (size_t)p * (psi_all ? N : 1) * DD;
    double* gAo = gLA + (size_t)p * N * DD;
    double* gbo = gLb + (size_t)p * N * D;
    const int vl = min(lane, D - 1);
    const bool jumps = jumpS && obs_flag;
    const double* jSp = jumps ? jumpS + (size_t)p * jumpS_stride : nullptr;
    const double h = 0.5 * dt;
    const int r1 = N > 1 ? N - 2 : 0, r2 = N > 2 ? N - 3 : 0;

    {
        const size_t last = (size_t)(N - 1);
        const double* A0 = Ap + last * DD;
        const double* A1 = Ap + (size_t)r1 * DD;
        const double* S0 = Sp + last * DD;
        double* P0 = Po + (psi_all ? last : 0) * DD;
        const bool store_last = psi_all || N == 1;
        for (int e = tid; e < DD; e += M40FH::NT) {
            const int row = e / D, col = e % D;
            lds[M40FH::OFF_AT + col * LD + row] = A0[e];
            lds[M40FH::OFF_AT + MS + col * LD + row] = RK2 ? 0.5 * (A0[e] + A1[e]) : A1[e];
            SG[row * LD + col] = S0[e];
            const double pv = Psi_last ? Psi_last[(size_t)p * DD + e] : 0.0;
            Pb[row * LD + col] = pv;
            if (store_last) P0[e] = pv;
        }
        if (tid < 48) mx[tid] = mp[last * D + (tid + D - 2) % D];
        if (tid >= 64 && tid < 64 + D) siv[tid - 64] = SigInv[(size_t)p * sig_stride + (size_t)(tid - 64) * (D + 1)];
    }

    // Every role passes four barriers per row t = N-1 .. 0; `cur` is the buffer of A(t)' (Heun: the
    // two buffers alternate; RK2: always the first).
    if (wid < 3) {
        // No contraction into fused multiply-adds in the recurrence (this block and the next): the
        // compiler decides that per unrolled copy, entries (i, j) and (j, i) live in different copies,
        // and the separate scan kernels, whose bits these wavefronts reproduce, do not contract either.
#pragma clang fp contract(off)
        // ---- matrix wavefronts of the recurrence ------------------------------------------
        __builtin_amdgcn_s_setprio(3);
        double s[3][4], Gt[3][4];
        {
            const M40Lane L(wid, lane);
            M40_FOR(c, reg) {
                s[c][reg] = Psi_last ? Psi_last[(size_t)p * DD + L.g[c][reg]] : 0.0;
                Gt[c][reg] = gS[(size_t)(N - 1) * DD + L.g[c][reg]];
            }
        }
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            const double* At0 = lds + M40FH::OFF_AT + cur * MS;
            const double* At1 = lds + M40FH::OFF_AT + (cur ^ 1) * MS;
            mfma_f64x4 acc[3];
            // The entry offsets are recomputed in each half of a stage (from a lane index the
            // compiler cannot see through) instead of occupying 36 registers for the whole scan.
            int lane_o = lane;
            __syncthreads();                   // Psi[t], A[t]' are in LDS
            if (t >= 1) {
                m40_rows_x_rows(Pb, At0, wid, lane, acc);
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) Mt[L.l[c][reg]] = acc[c][reg];
            }
            __syncthreads();                   // Psi A complete
            if (t >= 1) {
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) {
                    const double mt = Mt[L.lt[c][reg]];
                    const double k1 = -Gt[c][reg] + (acc[c][reg] + mt);   // core/bwd_ode.py:43
                    if (RK2) {
                        Pb[L.l[c][reg]] = s[c][reg] - h * k1;             // (:345)
                    } else {
                        Pb[L.l[c][reg]] = s[c][reg] - k1 * dt;            // predictor (:219)
                        s[c][reg] = s[c][reg] - h * k1;
                    }
                }
            }
            __syncthreads();                   // predictor, A[t-1]', dEsde_dS[t-1] are in LDS
            bool on = false;
            if (t >= 1) {
                on = jumps && obs_flag[t - 1];
                m40_rows_x_rows(Pb, At1, wid, lane, acc);
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) {
                    Mt[L.l[c][reg]] = acc[c][reg];
                    const double gp = SG[L.l[c][reg]];
                    acc[c][reg] = RK2 ? 0.5 * (gp + Gt[c][reg]) : gp;     // the G of stage 1 (acc is in Mt now)
                    Gt[c][reg] = gp;
                }
            }
            __syncthreads();                   // X A complete; the LDS inputs are no longer read
            if (t >= 1) {
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) {
                    const double mt = Mt[L.lt[c][reg]];
                    const double k2 = -acc[c][reg] + (Mt[L.l[c][reg]] + mt);
                    double sn = RK2 ? s[c][reg] - k2 * dt                 // (:352)
                                    : s[c][reg] - h * k2;                 // Psi - dt/2 (k1 + k2) (:226)
                    if (on) sn += jSp[L.g[c][reg]];   // observed rows only; uniform branch
                    s[c][reg] = sn;
                    Pb[L.l[c][reg]] = sn;
                }
            }
            if (!RK2) cur ^= 1;
        }
    } else if (wid == 3) {
#pragma clang fp contract(off)
        // ---- memory wavefront of the recurrence: A' in, lam, Psi(t) out ------------------------
        __builtin_amdgcn_s_setprio(2);
        double lv = lam_last ? lam_last[(size_t)p * D + vl] : 0.0;
        if (lane < D) {
            lamv[lane] = lv;
            lo[(size_t)(N - 1) * D + lane] = lv;
        }
        // Heun: preA holds A[t-2], which replaces A[t]' once stage 0 is over (that buffer is stage 1's
        // matrix of the next step).  RK2: preA holds A[t-1], which replaces A[t]' likewise; A[t-2] is then
        // requested into the same registers and, one stage later, averaged with the LDS image of A[t-1]'
        // into the buffer of the mean.
        double preA[2 * M40Rows::NQ];
        M40Rows::fetch(Ap + (size_t)(RK2 ? r1 : r2) * DD, lane, preA);
        double g0 = gm[(size_t)(N - 1) * D + vl], g1 = gm[(size_t)r1 * D + vl], gpre = gm[(size_t)r2 * D + vl];
        double jcur = jmp[(size_t)r1 * D + vl], jpre = jmp[(size_t)r2 * D + vl];
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            double* At0 = lds + M40FH::OFF_AT + cur * MS;
            double* At1 = lds + M40FH::OFF_AT + (cur ^ 1) * MS;
            __syncthreads();                   // Psi[t], A[t]', lam[t] are in LDS
            if (t < N - 1 && (psi_all || t == 0)) M40Rows::write_out(Pb, lane, Po + (size_t)t * DD);
            double f1 = 0.0;
            if (t >= 1 && lane < D) {
                double d0 = 0.0, d1 = 0.0;     // (A lam)[lane] = sum_q At[q][lane] lam[q]
#pragma unroll 5
                for (int k = 0; k < D; k += 2) {
                    d0 = fma(At0[k * LD + lane], lamv[k], d0);
                    d1 = fma(At0[(k + 1) * LD + lane], lamv[k + 1], d1);
                }
                f1 = -g0 + (d0 + d1);                                   // core/bwd_ode.py:42
            }
            __syncthreads();                   // A[t]' and lam[t] are no longer read
            if (t >= 1) {
                if (lane < D) lamv[lane] = RK2 ? lv - h * f1 : lv - f1 * dt;    // stage argument (:344, :218)
                M40Rows::commit<true>(preA, lane, At0);
                const int tp = RK2 ? ((t >= 2) ? t - 2 : 0) : ((t >= 3) ? t - 3 : 0);
                M40Rows::fetch(Ap + (size_t)tp * DD, lane, preA);
            }
            __syncthreads();                   // stage argument and stage matrix are in LDS
            double lnew = 0.0;
            if (t >= 1 && lane < D) {
                double d0 = 0.0, d1 = 0.0;
#pragma unroll 5
                for (int k = 0; k < D; k += 2) {
                    d0 = fma(At1[k * LD + lane], lamv[k], d0);
                    d1 = fma(At1[(k + 1) * LD + lane], lamv[k + 1], d1);
                }
                if (RK2) {
                    const double f2 = -0.5 * (g1 + g0) + (d0 + d1);
                    lnew = lv - f2 * dt + jcur;                         // (:351), jump (:121-124)
                } else {
                    const double f2 = -g1 + (d0 + d1);
                    lnew = lv - h * (f1 + f2) + jcur;                   // (:225), jump
                }
                lo[(size_t)(t - 1) * D + lane] = lnew;
            }
            __syncthreads();
            if (t >= 1) {
                if (lane < D) lamv[lane] = lnew;
                lv = lnew;
                // RK2: the mean of A[t-1] (now in the first buffer) and A[t-2] (registers) for the next step
                if (RK2) m40rows_mid_transposed(At0, preA, lane, At1);
                g0 = g1;
                g1 = gpre;
                jcur = jpre;
                const int tp = (t >= 3) ? t - 3 : 0;
                gpre = gm[(size_t)tp * D + vl];
                jpre = jmp[(size_t)tp * D + vl];
            }
            if (!RK2) cur ^= 1;
        }
    } else if (wid < 7) {
        // ---- gradient wavefronts: tile row r of X S -------------------------------------------
        const int r = wid - 4;
        const int g = lane >> 4, q = lane & 15;
        const int i = min(16 * r + q, D - 1);                 // matrix row this lane feeds as operand row q
        const int k0 = 10 * g;
        // columns of row i of <df/dx> that are not zero: i, i + 1, i - 1, i - 2 on the ring
        const int jf1 = (i + 1) % D, jb1 = (i + D - 1) % D, jb2 = (i + D - 2) % D;
        // the three gradient wavefronts stage S(t) themselves, a third each, one row ahead
        double preS[2 * M40Third::NQ];
        M40Third::fetch(Sp + (size_t)r1 * DD, r, lane, preS);
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            const double* At = lds + M40FH::OFF_AT + cur * MS;
            __syncthreads();                   // row t: A', Psi, S, m are in LDS
            double x[10];
            mfma_f64x4 z[3];
            {
                const double si = siv[i];
                const double m_b1 = mx[i + 1], m_f1 = mx[i + 3], m_b2 = mx[i];      // m[i-1], m[i+1], m[i-2]
                const double e_f1 = m_b1, e_b2 = -m_b1, e_b1 = m_f1 - m_b2;
                const double* prow = Pb + i * LD + k0;
#pragma unroll
                for (int hh = 0; hh < 5; ++hh) {
                    const double2 pv = *reinterpret_cast<const double2*>(prow + 2 * hh);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kk = 2 * hh + u, k = k0 + kk;
                        const double av = At[k * LD + i];
                        double ed = 0.0;                              // dynamics/sys_lorenz_96_ut.py:322-325
                        ed = (k == i) ? -1.0 : ed;
                        ed = (k == jf1) ? e_f1 : ed;
                        ed = (k == jb2) ? e_b2 : ed;
                        ed = (k == jb1) ? e_b1 : ed;
                        x[kk] = si * (ed + av) - 2.0 * (u ? pv.y : pv.x);
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double* srow = SG + min(16 * c + q, D - 1) * LD + k0;
                    double sb[10];
#pragma unroll
                    for (int hh = 0; hh < 5; ++hh) {
                        const double2 v = *reinterpret_cast<const double2*>(srow + 2 * hh);
                        sb[2 * hh] = v.x;
                        sb[2 * hh + 1] = v.y;
                    }
                    z[c] = mfma_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < 10; ++k) z[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[k], sb[k], z[c], 0, 0, 0);
                }
            }
            __syncthreads();                   // the LDS images of row t are no longer read; s1 + s2 is in LDS
            {
                const double s12 = sc[0];
                double* out = gAo + (size_t)t * DD;
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(r, lane_o);
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) out[L.g[c][reg]] = dt * (z[c][reg] - s12);   // core/gradL_Ab.py:170, :176
            }
            __syncthreads();                   // stage 1: the buffer holds dEsde_dS[t-1]
            __syncthreads();                   // ... and is free again
            if (t >= 1) {
                M40Third::commit(preS, r, lane, SG);
                M40Third::fetch(Sp + (size_t)(t >= 2 ? t - 2 : 0) * DD, r, lane, preS);
            }
            if (!RK2) cur ^= 1;
        }
    } else {
        // ---- second memory wavefront: dEsde_dS, m, <f>, b in; dL/db and s1 + s2 -------------------
        double preG[2 * M40Rows::NQ];          // dEsde_dS[t-1], put into LDS for stage 1
        M40Rows::fetch(gS + (size_t)r1 * DD, lane, preG);
        double ef_cur = efp[(size_t)(N - 1) * D + vl], b_cur = bp[(size_t)(N - 1) * D + vl];
        double ef_nxt = efp[(size_t)r1 * D + vl], b_nxt = bp[(size_t)r1 * D + vl];
        // the next row of m for the ring image: entry `lane` of mx (lanes 0 .. 47)
        const int mxi = (min(lane, 47) + D - 2) % D;
        double m_nxt = mp[(size_t)r1 * D + mxi];
        int cur = 0;
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            const double* At = lds + M40FH::OFF_AT + cur * MS;
            __syncthreads();                   // row t: A', lam, m are in LDS
            {
                double s12 = 0.0;
                if (lane < D) {
                    double am0 = 0.0, am1 = 0.0;       // (A m)[lane] = sum_k At[k][lane] m[k]
#pragma unroll 5
                    for (int k = 0; k < D; k += 2) {
                        am0 = fma(At[k * LD + lane], mx[k + 2], am0);
                        am1 = fma(At[(k + 1) * LD + lane], mx[k + 3], am1);
                    }
                    const double ml = mx[lane + 2], ll = lamv[lane];
                    const double u = -ef_cur - (am0 + am1) + b_cur;     // core/gradL_Ab.py:164
                    const double d = u * siv[lane];
                    gbo[(size_t)t * D + lane] = dt * (d + ll);          // :173, :177
                    s12 = d * ml + ll * ml;                             // :167, :170
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s12 += __shfl_xor(s12, off, 64);
                if (lane == 0) sc[0] = s12;
            }
            __syncthreads();                   // S[t], m[t] no longer read
            if (t >= 1) {
                M40Rows::commit<false>(preG, lane, SG);
                if (lane < 48) mx[lane] = m_nxt;
                ef_cur = ef_nxt;
                b_cur = b_nxt;
                const int tp = (t >= 2) ? t - 2 : 0;
                M40Rows::fetch(gS + (size_t)tp * DD, lane, preG);
                ef_nxt = efp[(size_t)tp * D + vl];
                b_nxt = bp[(size_t)tp * D + vl];
                m_nxt = mp[(size_t)tp * D + mxi];
            }
            __syncthreads();
            __syncthreads();
            if (!RK2) cur ^= 1;
        }
    }
}
