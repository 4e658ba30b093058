// Backward Euler scan at D = 40 with symmetric state, fused with the gradient assembly.
//
// One 512-thread workgroup per problem walks the time rows t = N-1 .. 0.  The first four
// wavefronts are those of bwd_euler_sym_mfma40_kernel (three matrix wavefronts advance
// Psi, the fourth stages A(t)' and dEsde_dS(t), integrates lam and writes Psi(t) out).  While
// row t is in LDS anyway -- A(t)', Psi(t), lam(t) -- four more wavefronts form the gradient
// of that row (core/gradL_Ab.py:156-177):
//
//   dL/dA(t) = dt ( [diag(SigInv) (<df/dx>(t) + A(t)) - 2 Psi(t)] S(t) - s1 - s2 )
//   dL/db(t) = dt ( db(t) + lam(t) ),   db = diag(SigInv) (-<f>(t) - A(t) m(t) + b(t)),
//   s1 = db . m(t),  s2 = lam(t) . m(t)          (the reference's scalar inner products)
//
// with <df/dx> of Lorenz-96 derived from the mean (dynamics/sys_lorenz_96_ut.py:293-330).
// Wavefronts 4..6 own the three tile rows of the 48 x 48 padded product X S on the matrix
// cores (X = diag(SigInv)(<df/dx> + A) - 2 Psi is built in registers from the LDS images, S(t)
// is read as rows: it is symmetric, and they stage it themselves one row ahead); wavefront 7
// streams m(t), <f>(t), b(t) in (and dEsde_dS(t) for the recurrence, which leaves the fourth
// wavefront A(t)' only) and forms db, dL/db and the scalar s1 + s2.  The
// recurrence's wavefronts run at a higher priority, so the gradient products fill the matrix
// cores behind them.
//
// Compared with the scan followed by the assembly kernel, A(t) and Psi(t) are not read a
// second time from HBM: 8 (5 D^2 + 7 D) = 66 240 bytes per (problem, step) instead of
// 39 360 + 53 120.
//
// Included by scan.hip after scan_mfma.h, inside its anonymous namespace.

struct M40F {
    static constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD;
    static constexpr int NT = 512;
    static constexpr int OFF_LAM = 5 * MS;          // lam(t)
    static constexpr int OFF_MX = OFF_LAM + D;      // m(t) on the ring: mx[k] = m[(k - 2) mod D], k = 0 .. 47
    static constexpr int OFF_SI = OFF_MX + 48;      // diagonal of SigInv
    static constexpr int OFF_SC = OFF_SI + D;       // s1 + s2 of the current row
    static constexpr int OFF_S2 = OFF_SC + 2;       // the second image of S: rows alternate between the two
    static constexpr int LDS_DOUBLES = OFF_S2 + MS; // 81 680 bytes: two workgroups fill the 160 KB of a compute unit
};

// One wavefront's traffic for a whole 40 x 40 matrix in 16-byte pieces, three matrix rows per
// instruction: instruction i moves rows 3 i .. 3 i + 2, lane l (< 60) the piece (row 3 i + l / 20,
// column pair l % 20), so that every address is one per-lane base plus a compile-time offset
// (no address registers per instruction).  The last instruction has row 39 only; the other
// lanes repeat its pieces (same values to the same addresses).  Lanes 60 .. 63 repeat lane 59.
struct M40Rows {
    static constexpr int NQ = 14;
    static constexpr int D = M40::D, LD = M40::LD;
    static __device__ __forceinline__ void split(int lane, int& lr, int& lc) {
        const int l60 = min(lane, 59);
        lr = (l60 >= 20) + (l60 >= 40);
        lc = l60 - 20 * lr;
    }
    static __device__ __forceinline__ void fetch(const double* __restrict__ src, int lane, double (&pre)[2 * NQ]) {
        int lr, lc;
        split(lane, lr, lc);
        const double2* s2 = reinterpret_cast<const double2*>(src) + lr * (D / 2) + lc;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const double2 v = vgpa::ld_once((i < NQ - 1) ? s2 + i * 3 * (D / 2) : s2 + i * 3 * (D / 2) - lr * (D / 2));
            pre[2 * i] = v.x;
            pre[2 * i + 1] = v.y;
        }
    }
    template <bool TRANSPOSE>
    static __device__ __forceinline__ void commit(const double (&pre)[2 * NQ], int lane, double* dst) {
        int lr, lc;
        split(lane, lr, lc);
        if (TRANSPOSE) {
            double* d0 = dst + 2 * lc * LD + lr;           // element (row, col) goes to [col][row]
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                double* d = (i < NQ - 1) ? d0 + 3 * i : d0 + 3 * i - lr;
                d[0] = pre[2 * i];
                d[LD] = pre[2 * i + 1];
            }
        } else {
            double* d0 = dst + lr * LD + 2 * lc;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                double* d = (i < NQ - 1) ? d0 + 3 * i * LD : d0 + 3 * i * LD - lr * LD;
                *reinterpret_cast<double2*>(d) = make_double2(pre[2 * i], pre[2 * i + 1]);
            }
        }
    }
    static __device__ __forceinline__ void write_out(const double* src, int lane, double* __restrict__ dst) {
        int lr, lc;
        split(lane, lr, lc);
        const double* s0 = src + lr * LD + 2 * lc;
        double2* d2 = reinterpret_cast<double2*>(dst) + lr * (D / 2) + lc;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const double* sp = (i < NQ - 1) ? s0 + 3 * i * LD : s0 + 3 * i * LD - lr * LD;
            double2* dp = (i < NQ - 1) ? d2 + i * 3 * (D / 2) : d2 + i * 3 * (D / 2) - lr * (D / 2);
            vgpa::st_once(dp, *reinterpret_cast<const double2*>(sp));
        }
    }
};

// A third of a 40 x 40 matrix in 16-byte pieces, for wavefront r of three: pieces
// lane + 64 (r + 3 q), q = 0 .. 4 (the 800 pieces are 12.5 groups of 64).
struct M40Third {
    static constexpr int NQ = 5;
    static __device__ __forceinline__ int piece(int r, int lane, int q) { return lane + 64 * (r + 3 * q); }
    static __device__ __forceinline__ void fetch(const double* __restrict__ src, int r, int lane, double (&pre)[2 * NQ]) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double2 v = vgpa::ld_once(s2 + min(piece(r, lane, q), M40::DD / 2 - 1));
            pre[2 * q] = v.x;
            pre[2 * q + 1] = v.y;
        }
    }
    static __device__ __forceinline__ void commit(const double (&pre)[2 * NQ], int r, int lane, double* dst) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pc = piece(r, lane, q);
            if (pc < M40::DD / 2) {
                const int e = 2 * pc;
                *reinterpret_cast<double2*>(dst + (e / M40::D) * M40::LD + (e % M40::D)) = make_double2(pre[2 * q], pre[2 * q + 1]);
            }
        }
    }
};

// PACKED_S, PACKED_G, PACKED_P: S, dEsde_dS, Psi_out are in the packed symmetric layout (sym_rows.h), 1152 doubles a
// row instead of 1600; A, dL/dA and the terminal state Psi_last stay dense.
template <bool PACKED_S, bool PACKED_G, bool PACKED_P>
__global__ void __launch_bounds__(M40F::NT, 4)
bwd_grad_euler_sym_mfma40_kernel(int N, const double* __restrict__ A, const double* __restrict__ bvec,
                                 const double* __restrict__ mvec, const double* __restrict__ Smat,
                                 const double* __restrict__ dEdm, const double* __restrict__ dEdS,
                                 const double* __restrict__ jm, const int* __restrict__ obs_flag,
                                 const double* __restrict__ jumpS, int jumpS_stride,
                                 const double* __restrict__ Efv, const double* __restrict__ SigInv, int sig_stride,
                                 double dt, const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                                 double* __restrict__ lam_out, double* __restrict__ Psi_out, int psi_all,
                                 double* __restrict__ gLA, double* __restrict__ gLb) {
    constexpr int D = M40::D, LD = M40::LD, MS = M40::MS, DD = M40::DD;
    extern __shared__ double lds[];
    double* Pb = lds;             // Psi[t]
    double* At = lds + MS;        // A[t] transposed
    double* Mt = lds + 2 * MS;    // Psi A, read back transposed
    double* Gb = lds + 3 * MS;    // dEsde_dS[t]
    double* Sb = lds + 4 * MS;    // S[t]
    double* lamv = lds + M40F::OFF_LAM;
    double* mx = lds + M40F::OFF_MX;
    double* siv = lds + M40F::OFF_SI;
    double* sc = lds + M40F::OFF_SC;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    constexpr int SS = PACKED_S ? M40Packed::SIZE : DD;     // doubles per stored S(t), dEsde_dS(t), Psi(t)
    constexpr int SG = PACKED_G ? M40Packed::SIZE : DD;
    constexpr int SP = PACKED_P ? M40Packed::SIZE : DD;
    const double* Sp = Smat + (size_t)p * N * SS;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * SG;
    const double* jmp = jm + (size_t)p * N * D;
    const double* bp = bvec + (size_t)p * N * D;
    const double* mp = mvec + (size_t)p * N * D;
    const double* efp = Efv + (size_t)p * N * D;
    double* lo = lam_out + (size_t)p * N * D;
    // psi_all == 0: Psi_out is (P, 1, D, D) and receives row 0 only
    double* Po = Psi_out + (size_t)p * (psi_all ? N : 1) * SP;
    double* gAo = gLA + (size_t)p * N * DD;
    double* gbo = gLb + (size_t)p * N * D;
    const int vl = min(lane, D - 1);
    const bool jumps = jumpS && obs_flag;
    const double* jSp = jumps ? jumpS + (size_t)p * jumpS_stride : nullptr;

    {
        const size_t last = (size_t)(N - 1);
        const double* A0 = Ap + last * DD;
        const double* G0 = gS + last * SG;
        const double* S0 = Sp + last * SS;
        double* P0 = Po + (psi_all ? last : 0) * SP;
        const bool store_last = psi_all || N == 1;
        for (int e = tid; e < DD; e += M40F::NT) {
            const int row = e / D, col = e % D;
            At[col * LD + row] = A0[e];
            if (!PACKED_G) Gb[row * LD + col] = G0[e];
            if (!PACKED_S) Sb[row * LD + col] = S0[e];
            const double pv = Psi_last ? Psi_last[(size_t)p * DD + e] : 0.0;
            Pb[row * LD + col] = pv;
            if (!PACKED_P && store_last) P0[e] = pv;
        }
        if (PACKED_G) M40Packed::load_image(G0, tid, M40F::NT, Gb);
        if (PACKED_S) M40Packed::load_image(S0, tid, M40F::NT, Sb);
        if (PACKED_P && store_last) M40Packed::pack_dense(Psi_last ? Psi_last + (size_t)p * DD : nullptr, tid, M40F::NT, P0);
        if (tid < 48) mx[tid] = mp[last * D + (tid + D - 2) % D];
        if (tid >= 64 && tid < 64 + D) siv[tid - 64] = SigInv[(size_t)p * sig_stride + (size_t)(tid - 64) * (D + 1)];
    }

    if (wid < 3) {
        // ---- matrix wavefronts of the recurrence ------------------------------------------
        __builtin_amdgcn_s_setprio(3);
        double s[3][4];
        {
            const M40Lane L(wid, lane);
            M40_FOR(c, reg) s[c][reg] = Psi_last ? Psi_last[(size_t)p * DD + L.g[c][reg]] : 0.0;
        }
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            __syncthreads();                   // Psi[t], A[t]', dEsde_dS[t] are in LDS
            M40F_STAMP(0);
            mfma_f64x4 acc[3];
            double Gt[3][4];
            bool on = false;
            // The entry offsets are recomputed in each half of a step (from a lane index the
            // compiler cannot see through) instead of occupying 36 registers for the whole scan.
            int lane_o = lane;
            if (t >= 1) {
                on = jumps && obs_flag[t - 1];
                m40_rows_x_rows(Pb, At, wid, lane, acc);
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) {
                    Mt[L.l[c][reg]] = acc[c][reg];
                    Gt[c][reg] = Gb[L.l[c][reg]];
                }
            }
            M40F_STAMP(1);
            __syncthreads();                   // Psi A complete; the LDS inputs are no longer read
            M40F_STAMP(2);
            if (t >= 1) {
                asm volatile("" : "+v"(lane_o));
                const M40Lane L(wid, lane_o);
                M40_FOR(c, reg) {
                    const double mt = Mt[L.lt[c][reg]];
                    const double k1 = -Gt[c][reg] + (acc[c][reg] + mt);   // core/bwd_ode.py:43
                    double sn = s[c][reg] - k1 * dt;
                    if (on) sn += jSp[L.g[c][reg]];   // observed rows only (:122, :127); uniform branch
                    s[c][reg] = sn;
                    Pb[L.l[c][reg]] = sn;
                }
            }
            M40F_STAMP(3);
        }
    } else if (wid == 3) {
        // ---- memory wavefront of the recurrence: A' in, lam, Psi(t) out ------------------------
        __builtin_amdgcn_s_setprio(2);
        double lv = lam_last ? lam_last[(size_t)p * D + vl] : 0.0;
        double gmt = gm[(size_t)(N - 1) * D + vl];
        if (lane < D) {
            lamv[lane] = lv;
            lo[(size_t)(N - 1) * D + lane] = lv;
        }
        double preA[2 * M40Rows::NQ];
        const int r1 = N > 1 ? N - 2 : 0;
        M40Rows::fetch(Ap + (size_t)r1 * DD, lane, preA);
        double gnext = gm[(size_t)r1 * D + vl];
        double jnext = jmp[(size_t)r1 * D + vl];
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            __syncthreads();                   // Psi[t], A[t]', lam[t] are in LDS
            M40F_STAMP(0);
            if (t < N - 1 && (psi_all || t == 0)) {
                if (PACKED_P) M40Packed(lane).write_out(Pb, Po + (size_t)t * SP);
                else M40Rows::write_out(Pb, lane, Po + (size_t)t * DD);
            }
            double lnew = 0.0;
            if (t >= 1 && lane < D) {
                double d0 = 0.0, d1 = 0.0;     // (A lam)[lane] = sum_q At[q][lane] lam[q]
#pragma unroll 5
                for (int k = 0; k < D; k += 2) {
                    d0 = fma(At[k * LD + lane], lamv[k], d0);
                    d1 = fma(At[(k + 1) * LD + lane], lamv[k + 1], d1);
                }
                const double f1 = -gmt + (d0 + d1);                     // core/bwd_ode.py:42
                lnew = lv - f1 * dt + jnext;                            // core/bwd_ode.py:121-124
                lo[(size_t)(t - 1) * D + lane] = lnew;
            }
            M40F_STAMP(1);
            __syncthreads();                   // Psi[t], A[t]', dEsde_dS[t], lam[t] no longer read
            M40F_STAMP(2);
            if (t >= 1) {
                if (lane < D) lamv[lane] = lnew;
                lv = lnew;
                gmt = gnext;
                M40Rows::commit<true>(preA, lane, At);
                M40F_STAMP(4);
                const int tp = (t >= 2) ? t - 2 : 0;
                M40Rows::fetch(Ap + (size_t)tp * DD, lane, preA);
                gnext = gm[(size_t)tp * D + vl];
                jnext = jmp[(size_t)tp * D + vl];
            }
            M40F_STAMP(3);
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
        double preS[PACKED_S ? 6 : 2 * M40Third::NQ];
        if constexpr (PACKED_S) M40Packed(lane).fetch_third(Sp + (size_t)(N > 1 ? N - 2 : 0) * SS, r, preS);
        else M40Third::fetch(Sp + (size_t)(N > 1 ? N - 2 : 0) * DD, r, lane, preS);
        // S(t) has two LDS images that take turns: the products of row t read one of them AFTER the second barrier
        // of the step, while the row for the next step is committed to the other.  Between the barriers these
        // wavefronts only collect their left operand from the images that are rewritten afterwards (A', Psi, m): the
        // matrix cores are then theirs while the recurrence's wavefronts update Psi, instead of being shared with
        // them before the barrier.
        double* Scur = Sb;
        double* Snext = lds + M40F::OFF_S2;
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            __syncthreads();                   // row t: A', Psi, m are in LDS (and S, in its own image)
            M40F_STAMP(0);
            double x[10];
            mfma_f64x4 z[3];
            {
                const double si = siv[i];
                const double m_b1 = mx[i + 1], m_f1 = mx[i + 3], m_b2 = mx[i];      // m[i-1], m[i+1], m[i-2]
                const double e_f1 = m_b1, e_b2 = -m_b1, e_b1 = m_f1 - m_b2;
                const double* prow = Pb + i * LD + k0;
#pragma unroll
                for (int h = 0; h < 5; ++h) {
                    const double2 pv = *reinterpret_cast<const double2*>(prow + 2 * h);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kk = 2 * h + u, k = k0 + kk;
                        const double av = At[k * LD + i];
                        double ed = 0.0;                              // dynamics/sys_lorenz_96_ut.py:322-325
                        ed = (k == i) ? -1.0 : ed;
                        ed = (k == jf1) ? e_f1 : ed;
                        ed = (k == jb2) ? e_b2 : ed;
                        ed = (k == jb1) ? e_b1 : ed;
                        x[kk] = si * (ed + av) - 2.0 * (u ? pv.y : pv.x);
                    }
                }
            }
            M40F_STAMP(1);
            __syncthreads();                   // the LDS images of row t (but S) are no longer read; s1 + s2 is in LDS
            M40F_STAMP(2);
            {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double* srow = Scur + min(16 * c + q, D - 1) * LD + k0;
                    double sb[10];
#pragma unroll
                    for (int h = 0; h < 5; ++h) {
                        const double2 v = *reinterpret_cast<const double2*>(srow + 2 * h);
                        sb[2 * h] = v.x;
                        sb[2 * h + 1] = v.y;
                    }
                    z[c] = mfma_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < 10; ++k) z[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[k], sb[k], z[c], 0, 0, 0);
                }
            }
            // the next row of S goes to the image that nobody reads in this step (measured: committing it here, behind
            // the products, is faster than before the second barrier)
            if (t >= 1) {
                if constexpr (PACKED_S) {
                    const M40Packed K(lane);
                    K.commit_third(preS, r, Snext);
                    M40F_STAMP(4);
                    K.fetch_third(Sp + (size_t)(t >= 2 ? t - 2 : 0) * SS, r, preS);
                } else {
                    M40Third::commit(preS, r, lane, Snext);
                    M40F_STAMP(4);
                    M40Third::fetch(Sp + (size_t)(t >= 2 ? t - 2 : 0) * DD, r, lane, preS);
                }
            }
            {
                double* tmp = Scur;
                Scur = Snext;
                Snext = tmp;
            }
            const double s12 = sc[0];
            double* out = gAo + (size_t)t * DD;
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const M40Lane L(r, lane_o);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) out[L.g[c][reg]] = dt * (z[c][reg] - s12);   // core/gradL_Ab.py:170, :176
            M40F_STAMP(3);
        }
    } else {
        // ---- second memory wavefront: dEsde_dS, m, <f>, b in; dL/db and s1 + s2 -------------------
        double preG[2 * (PACKED_G ? M40Packed::NQ : M40Rows::NQ)];
        const int r1 = N > 1 ? N - 2 : 0;
        if constexpr (PACKED_G) M40Packed(lane).fetch(gS + (size_t)r1 * SG, preG);
        else M40Rows::fetch(gS + (size_t)r1 * DD, lane, preG);
        double ef_cur = efp[(size_t)(N - 1) * D + vl], b_cur = bp[(size_t)(N - 1) * D + vl];
        double ef_nxt = efp[(size_t)r1 * D + vl], b_nxt = bp[(size_t)r1 * D + vl];
        // the next row of m for the ring image: entry `lane` of mx (lanes 0 .. 47)
        const int mxi = (min(lane, 47) + D - 2) % D;
        double m_nxt = mp[(size_t)r1 * D + mxi];
#pragma unroll 1
        for (int t = N - 1; t >= 0; --t) {
            __syncthreads();                   // row t: A', lam, m are in LDS
            M40F_STAMP(0);
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
            M40F_STAMP(1);
            __syncthreads();                   // S[t], m[t] no longer read
            M40F_STAMP(2);
            if (t >= 1) {
                if constexpr (PACKED_G) M40Packed(lane).commit<true>(preG, Gb);
                else M40Rows::commit<false>(preG, lane, Gb);
                M40F_STAMP(4);
                if (lane < 48) mx[lane] = m_nxt;
                M40F_STAMP(5);
                ef_cur = ef_nxt;
                b_cur = b_nxt;
                const int tp = (t >= 2) ? t - 2 : 0;
                if constexpr (PACKED_G) M40Packed(lane).fetch(gS + (size_t)tp * SG, preG);
                else M40Rows::fetch(gS + (size_t)tp * DD, lane, preG);
                ef_nxt = efp[(size_t)tp * D + vl];
                b_nxt = bp[(size_t)tp * D + vl];
                m_nxt = mp[(size_t)tp * D + mxi];
            }
            M40F_STAMP(3);
        }
    }
}
