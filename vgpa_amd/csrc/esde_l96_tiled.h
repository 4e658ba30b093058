// Lorenz-96 unscented energy kernel, compile-time dimension: one 256-thread
// workgroup per (problem, time step), all D x D operands in LDS.
//
// Phases (block barriers between them unless noted):
//   1  load S(t) and A(t) (16-byte coalesced loads); S(t) and m go to LDS, A(t) waits in
//      registers until after phase 3 so that its latency hides behind the pivot chain
//   2  <f>, A m - b, optionally the dense <df/dx> (the covariance term before phase 3,
//      the rest after A(t) has been put into LDS)
//   3  blocked Cholesky S = L L' (block width 8).  Per block column: update by all
//      threads, barrier, then ONE wavefront factors the 8x8 diagonal block and inverts
//      the factor (wave-synchronous, cross-lane register reads), barrier, then all
//      threads solve the row panel below it, barrier.  The pivot recurrence
//      is the square-root-free one (one reciprocal per pivot on the critical
//      path); the square roots are taken off the chain.
//   4  G = A L with 2x4 register tiles and 16-byte LDS reads, then the +/-
//      sigma-point residuals of the tile; partial sums of v+-_i over d are
//      combined between lane pairs, then stored per tile pair
//   5  wavefront 0 reduces the partial sums -> Esde(t), u = v+ - v-, g;
//      CONCURRENTLY wavefronts 1..3 compute W = L^-1 by block columns
//      (wave-synchronous chains, no block barrier)
//   6  dEsde/dm = (w c / 2) W' u;  V = diag(g) W
//   7  dEsde/dS = W' V on the upper tiles, written to (i,j) and (j,i): the result
//      is bitwise symmetric
//
// HYPER adds the per-component <(f-g)^2> integrand (dSig_t), which costs one more
// LDS array and one resident workgroup per CU; the free-energy path runs without.
//
// Included by esde.hip inside its anonymous namespace.

template <int D, bool HYPER>
struct L96Tiled {
    static constexpr int LD = D + 2;          // even: rows stay 16-byte aligned
    static constexpr int MS = D * LD;
    static constexpr int NB = 8;              // Cholesky block width
    static constexpr int NBLK = D / NB;
    static constexpr int TI = 2, TJ = 4;      // register tile
    static constexpr int NTI = D / TI, NTJ = D / TJ;
    static constexpr int NPAIR = NTI / 2;
    static constexpr int NT = 256;
    static_assert(D % NB == 0, "D must be a multiple of 8");
    static_assert(NTI * NTJ <= NT, "one register tile per thread");
    // Tiles (ti, tj) of the symmetric output that touch the upper triangle, listed by
    // column tile: column tile tj has 2 tj + 2 of them (2 tj + 1 when clipped), so tile
    // number q sits in the column tile with tj (tj + 1) <= q.
    static constexpr int NUPPER = NTJ * (NTJ + 1) - ((2 * NTJ > NTI) ? (2 * NTJ - NTI) : 0);
    static constexpr int XB = 2 * NPAIR * D + (HYPER ? NTJ * D : 0);
    // Lm, Am | partial sums | 8x8 inverses | per-wave 8x8 scratch | vectors | scalars
    static constexpr int LDS_DOUBLES = 2 * MS + XB + NBLK * NB * NB + 4 * NB * NB + 6 * D + 8;
};

// 1/sqrt(p) and 1/p: hardware estimate plus Newton refinement.
__device__ inline double rsqrt_refined(double p) {
    double r = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}
__device__ inline double rcp_refined(double p) {
    double q = __builtin_amdgcn_rcp(p);
    double e = fma(-p, q, 1.0);
    q = fma(q, e, q);
    e = fma(-p, q, 1.0);
    q = fma(q, e, q);
    return q;
}


#ifdef VGPA_STAMPS
// Diagnostic build only: per-phase cycle counts of one workgroup, lane 0.
__device__ unsigned long long g_l96_stamps[32];
#define VGPA_STAMP(k) do { if (blockIdx.x == 60000 && threadIdx.x == 0) g_l96_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VGPA_STAMP(k) do { } while (0)
#endif

template <int D, bool HYPER>
__global__ void __launch_bounds__(256) esde_l96_ut_tiled_kernel(EsdeArgs a) {
    using C = L96Tiled<D, HYPER>;
    constexpr int LD = C::LD, MS = C::MS, NB = C::NB, NBLK = C::NBLK, NT = C::NT;
    constexpr int TI = C::TI, TJ = C::TJ, NTI = C::NTI, NTJ = C::NTJ, NPAIR = C::NPAIR;
    constexpr int DD = D * D;
    extern __shared__ double lds[];
    double* Lm = lds;                         // S -> L -> V = diag(g) W
    double* Am = lds + MS;                    // A -> W = L^-1
    double* Vp = lds + 2 * MS;                // [NPAIR][D] partial sums of v+
    double* Vm = Vp + NPAIR * D;              // [NPAIR][D] partial sums of v-
    double* Mb = Vm + NPAIR * D;              // [NTJ][D] partial sums of w (r+^2 + r-^2)  (HYPER)
    double* Tb = lds + 2 * MS + C::XB;        // [NBLK][8][8] inverses of the diagonal blocks of L
    double* Pw = Tb + NBLK * NB * NB;         // [4][8][8] per-wavefront scratch
    double* mv = Pw + 4 * NB * NB;
    double* basev = mv + D;                   // A m - b
    double* sIv = basev + D;
    double* uv = sIv + D;                     // v+ - v-
    double* gv = uv + D;                      // (v+ + v-)/8 - Esde/2
    double* a0v = gv + D;                     // squared centre-point residuals
    double* sc = a0v + D;                     // [0] = pivot failure flag
    const int tid = threadIdx.x;
    const int wid = tid >> 6, lane = tid & 63;
    const size_t e = blockIdx.x;              // p * N + t
    const int p = (int)(e / a.N);
    const double theta = a.theta[(size_t)p * a.theta_stride];

    VGPA_STAMP(0);
    // ---- phase 1: S(t) goes to LDS at once; A(t) stays in registers until the ---------
    // factorisation is done, so that its HBM latency hides behind the pivot chain
    constexpr int NQ = (DD / 2 + NT - 1) / NT;
    double apx[NQ], apy[NQ];
    {
        const double2* Ag = reinterpret_cast<const double2*>(a.A + e * DD);
        const double2* Sg = reinterpret_cast<const double2*>(a.S + e * DD);
        double spx[NQ], spy[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const double2 v = Sg[min(i * NT + tid, DD / 2 - 1)];
            spx[i] = v.x;
            spy[i] = v.y;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const double2 v = Ag[min(i * NT + tid, DD / 2 - 1)];
            apx[i] = v.x;
            apy[i] = v.y;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = i * NT + tid;
            if (q < DD / 2)
                *reinterpret_cast<double2*>(Lm + ((2 * q) / D) * LD + ((2 * q) % D)) = make_double2(spx[i], spy[i]);
        }
        if (tid < D) {
            mv[tid] = a.m[e * D + tid];
            sIv[tid] = a.sig_inv_diag[(size_t)p * a.sig_stride + tid];
        }
        if (tid == 0) sc[0] = 0.0;
    }
    __syncthreads();

    VGPA_STAMP(1);
    // ---- phase 2a: the covariance term of <f>, before S is overwritten by its factor ----
    double Cxx = 0.0;
    if (tid < D) {
        const int l = tid;
        const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + D - 2) % D;
        Cxx = Lm[f1 * LD + b1] - Lm[b2 * LD + b1];
    }
    __syncthreads();

    VGPA_STAMP(2);
    // ---- phase 3: blocked Cholesky ------------------------------------------
#pragma unroll 1
    for (int J = 0; J < NBLK; ++J) {
        const int c0 = J * NB;
        // (a) S[i][c0+c] -= sum_{k<c0} L[i][k] L[c0+c][k]   for rows i >= c0
        if (J > 0) {
            const int i = c0 + tid / NB, c = tid % NB;
            if (i < D) {
                const double* ri = Lm + i * LD;
                const double* rc = Lm + (c0 + c) * LD;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
                for (int k = 0; k < c0; k += 2) {
                    const double2 x = *reinterpret_cast<const double2*>(ri + k);
                    const double2 y = *reinterpret_cast<const double2*>(rc + k);
                    s0 = fma(x.x, y.x, s0);
                    s1 = fma(x.y, y.y, s1);
                }
                Lm[i * LD + c0 + c] -= (s0 + s1);
            }
            __syncthreads();
        }
        // (b) wavefront 0: factor the 8x8 diagonal block, invert its factor, solve the
        //     rows below.  Lane j (mod 8) holds column j of the symmetric block, read
        //     from the lower triangle; U = L' is built row by row.
        if (wid == 0) {
            const int j = lane & 7;
            double s[NB], piv[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) s[i] = Lm[(c0 + j) * LD + c0 + i];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                piv[k] = readlane_f64(s[k], k);
                ok = ok && (piv[k] > 0.0);
                const double tk = s[k] * rcp_refined(piv[k]);   // row k over its pivot
#pragma unroll
                for (int i = k + 1; i < NB; ++i) s[i] = fma(-readlane_f64(tk, i), s[k], s[i]);
            }
            double rinv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                rinv[k] = rsqrt_refined(piv[k]);
                s[k] *= rinv[k];                       // U[k][j]
            }
            // Y = U^-1 (upper), column j in this lane; T = Y' inverts the lower factor.
            // Terms are accumulated oldest-first so each row waits on one new value only.
            double y[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) y[i] = 0.0;
#pragma unroll
            for (int i = NB - 1; i >= 0; --i) {
                double acc = 0.0;
#pragma unroll
                for (int k = NB - 1; k > i; --k) acc = fma(readlane_f64(s[i], k), y[k], acc);
                y[i] = (j == i) ? rinv[i] : -rinv[i] * acc;
            }
            if (lane < NB) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (i <= j) Lm[(c0 + j) * LD + c0 + i] = s[i];          // L[j][i] = U[i][j]
                    Tb[J * NB * NB + j * NB + i] = (i <= j) ? y[i] : 0.0;     // T[j][i] = Y[i][j]
                }
                if (!ok && lane == 0) sc[0] = 1.0;
            }
        }
        __syncthreads();
        if (sc[0] != 0.0) break;   // uniform: every thread reads the same flag
        // (c) row panel, all threads: L[i][c0..c0+7] = S'[i][c0..c0+7] T'  for rows i >= c0 + 8.
        //     The eight threads of a row sit in one wavefront, so every one of them has
        //     read the row before any of them overwrites its entry.
        if (c0 + NB < D) {
            const int i = c0 + NB + tid / NB, c = tid % NB;
            if (i < D) {
                const double* T = Tb + J * NB * NB + c * NB;   // zero beyond the diagonal
                const double* row = Lm + i * LD + c0;
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int k = 0; k < NB; k += 2) {
                    const double2 x = *reinterpret_cast<const double2*>(row + k);
                    const double2 t = *reinterpret_cast<const double2*>(T + k);
                    acc0 = fma(x.x, t.x, acc0);
                    acc1 = fma(x.y, t.y, acc1);
                }
                wave_lds_fence();
                Lm[i * LD + c0 + c] = acc0 + acc1;
            }
            __syncthreads();
        }
    }
    VGPA_STAMP(3);
    // ---- A(t) into LDS, then phase 2b: <f>, A m - b, optionally the dense <df/dx> -------
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = i * NT + tid;
        if (q < DD / 2)
            *reinterpret_cast<double2*>(Am + ((2 * q) / D) * LD + ((2 * q) % D)) = make_double2(apx[i], apy[i]);
    }
    __syncthreads();
    if (tid < D) {
        const int l = tid;
        const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + D - 2) % D;
        double am0 = 0.0, am1 = 0.0;
#pragma unroll 5
        for (int k = 0; k < D; k += 2) {
            const double2 av = *reinterpret_cast<const double2*>(Am + l * LD + k);
            am0 = fma(av.x, mv[k], am0);
            am1 = fma(av.y, mv[k + 1], am1);
        }
        const double base = (am0 + am1) - a.b[e * D + l];
        basev[l] = base;
        const double f0 = (mv[f1] - mv[b2]) * mv[b1] - mv[l] + theta;   // drift at the mean
        const double Ef = Cxx + f0;
        a.Ef[e * D + l] = Ef;
        if (a.dth_t) a.dth_t[e * D + l] = Ef + base;
        const double r0 = f0 + base;
        a0v[l] = r0 * r0;
    }
    if (a.Edf) {
        // Jacobian of the drift at the mean; row i is filled in the order
        // [i] = -1, [i+1] = m[i-1], [i-2] = -m[i-1], [i-1] = m[i+1] - m[i-2].
        double* E = a.Edf + e * DD;
#pragma unroll
        for (int q0 = 0; q0 < DD; q0 += NT) {
            const int q = q0 + tid;
            if (q < DD) {
                const int i = q / D, j = q % D;
                const int jf1 = (i + 1) % D, jb1 = (i + D - 1) % D, jb2 = (i + D - 2) % D;
                double v = 0.0;
                if (j == i) v = -1.0;
                if (j == jf1) v = mv[jb1];
                if (j == jb2) v = -mv[jb1];
                if (j == jb1) v = mv[jf1] - mv[jb2];
                E[q] = v;
            }
        }
    }
    if (sc[0] != 0.0) {
        // Not positive definite.  The reference falls back to diagonal sigma points,
        // which needs a general inverse of S: flag the step for the fallback pass that
        // follows this kernel and poison its outputs in case there is none.
        const double nan = __builtin_nan("");
        if (tid == 0) {
            a.Esde_t[e] = nan;
            if (a.status) a.status[e] = VGPA_ST_CHOL_FALLBACK | VGPA_ST_UNSUPPORTED;
        }
        if (tid < D) {
            a.dEdm[e * D + tid] = nan;
            if (a.dSig_t) a.dSig_t[e * D + tid] = nan;
        }
        for (int q = tid; q < DD; q += NT) a.dEdS[e * DD + q] = nan;
        return;
    }
    // strict upper triangle of L := 0 (column reads below need no masking)
#pragma unroll
    for (int q0 = 0; q0 < DD; q0 += NT) {
        const int q = q0 + tid;
        if (q < DD) {
            const int i = q / D, j = q % D;
            if (j > i) Lm[i * LD + j] = 0.0;
        }
    }
    __syncthreads();

    const double kappa = 1.05 * D;
    const double c2 = D + kappa;
    const double cs = sqrt(c2);
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);

    VGPA_STAMP(4);
    // ---- phase 4: G = A L and the sigma-point residuals ----------------------
    if (tid < NTI * NTJ) {
        const int ij = tid / NTI, di = tid % NTI;      // lanes 2k, 2k+1 share ij, adjacent di
        const int d0 = di * TI, i0 = ij * TJ;
        double G[TI][TJ] = {};
        const double* A0 = Am + d0 * LD;
        const double* A1 = A0 + LD;
#pragma unroll 2
        for (int j = i0; j < D; j += 2) {
            const double2 a0 = *reinterpret_cast<const double2*>(A0 + j);
            const double2 a1 = *reinterpret_cast<const double2*>(A1 + j);
            const double2 x0 = *reinterpret_cast<const double2*>(Lm + j * LD + i0);
            const double2 x1 = *reinterpret_cast<const double2*>(Lm + j * LD + i0 + 2);
            const double2 y0 = *reinterpret_cast<const double2*>(Lm + (j + 1) * LD + i0);
            const double2 y1 = *reinterpret_cast<const double2*>(Lm + (j + 1) * LD + i0 + 2);
            const double xs[TJ] = {x0.x, x0.y, x1.x, x1.y}, ys[TJ] = {y0.x, y0.y, y1.x, y1.y};
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                G[0][c] = fma(a0.x, xs[c], G[0][c]);
                G[1][c] = fma(a1.x, xs[c], G[1][c]);
            }
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                G[0][c] = fma(a0.y, ys[c], G[0][c]);
                G[1][c] = fma(a1.y, ys[c], G[1][c]);
            }
        }
        // rows d0-2 .. d0+2 (cyclic) of columns i0..i0+3 of c L, and the matching means
        int rr[5];
        double mm[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            rr[q] = (d0 + D - 2 + q) % D;
            mm[q] = mv[rr[q]];
        }
        double o[5][TJ];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double2 v0 = *reinterpret_cast<const double2*>(Lm + rr[q] * LD + i0);
            const double2 v1 = *reinterpret_cast<const double2*>(Lm + rr[q] * LD + i0 + 2);
            o[q][0] = cs * v0.x; o[q][1] = cs * v0.y; o[q][2] = cs * v1.x; o[q][3] = cs * v1.y;
        }
        const double bs[TI] = {basev[d0], basev[d0 + 1]};
        const double si[TI] = {sIv[d0], sIv[d0 + 1]};
        double mb[TI] = {0.0, 0.0};
        double vpc[TJ], vmc[TJ];
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            vpc[c] = 0.0;
            vmc[c] = 0.0;
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                // state index d = d0 + r sits at position r + 2 of the 5-row window
                const double xm2p = mm[r] + o[r][c], xm1p = mm[r + 1] + o[r + 1][c];
                const double x0p = mm[r + 2] + o[r + 2][c], xp1p = mm[r + 3] + o[r + 3][c];
                const double xm2m = mm[r] - o[r][c], xm1m = mm[r + 1] - o[r + 1][c];
                const double x0m = mm[r + 2] - o[r + 2][c], xp1m = mm[r + 3] - o[r + 3][c];
                const double fp = (xp1p - xm2p) * xm1p - x0p + theta;
                const double fm = (xp1m - xm2m) * xm1m - x0m + theta;
                const double cG = cs * G[r][c];
                const double rp = fp + bs[r] + cG, rm = fm + bs[r] - cG;
                const double rp2 = rp * rp, rm2 = rm * rm;
                vpc[c] = fma(si[r], rp2, vpc[c]);
                vmc[c] = fma(si[r], rm2, vmc[c]);
                if (HYPER) mb[r] += w1 * (rp2 + rm2);
            }
        }
        // combine the two d-tiles of a lane pair, even lane stores
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            vpc[c] += __shfl_xor(vpc[c], 1, 64);
            vmc[c] += __shfl_xor(vmc[c], 1, 64);
        }
        if ((di & 1) == 0) {
            double* vp = Vp + (di >> 1) * D + i0;
            double* vm = Vm + (di >> 1) * D + i0;
            *reinterpret_cast<double2*>(vp) = make_double2(vpc[0], vpc[1]);
            *reinterpret_cast<double2*>(vp + 2) = make_double2(vpc[2], vpc[3]);
            *reinterpret_cast<double2*>(vm) = make_double2(vmc[0], vmc[1]);
            *reinterpret_cast<double2*>(vm + 2) = make_double2(vmc[2], vmc[3]);
        }
        if (HYPER) {
#pragma unroll
            for (int r = 0; r < TI; ++r) Mb[ij * D + d0 + r] = mb[r];
        }
    }
    __syncthreads();

    VGPA_STAMP(5);
    // ---- phase 5: wavefront 0 reduces, wavefronts 1..3 invert L ----------------
    double* Wm = Am;
    if (wid == 0) {
        const int l = lane;
        double vpl = 0.0, vml = 0.0, a0 = 0.0;
        if (l < D) {
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                vpl += Vp[q * D + l];
                vml += Vm[q * D + l];
            }
            a0 = sIv[l] * a0v[l];
        }
        const double v0 = wave_sum(a0);
        const double Esde = 0.5 * (w0 * v0 + w1 * wave_sum(vpl + vml));
        if (l < D) {
            uv[l] = vpl - vml;
            gv[l] = 0.125 * (vpl + vml) - 0.5 * Esde;
            if (HYPER && a.dSig_t) {
                double mbar = w0 * a0v[l];
#pragma unroll
                for (int q = 0; q < NTJ; ++q) mbar += Mb[q * D + l];
                a.dSig_t[e * D + l] = mbar;
            }
        }
        if (l == 0) {
            a.Esde_t[e] = Esde;
            if (a.status) a.status[e] = 0;
        }
    } else {
        // W = L^-1, one block column per pass: W_JJ = T_J,
        // W_IJ = -T_I sum_{K=J}^{I-1} L_IK W_KJ.  Lane (r, c) owns entry (r, c) of the
        // current 8x8 block; the chain over I is wave-synchronous.
        const int r = lane >> 3, c = lane & 7;
        double* Pq = Pw + wid * NB * NB;
        for (int J = wid - 1; J < NBLK; J += 3) {
            const int c0 = J * NB;
            for (int i = lane; i < c0 * NB; i += 64) Wm[(i / NB) * LD + c0 + (i % NB)] = 0.0;   // blocks above
            Wm[(c0 + r) * LD + c0 + c] = Tb[J * NB * NB + r * NB + c];
            wave_lds_fence();
            for (int I = J + 1; I < NBLK; ++I) {
                const int r0 = I * NB;
                const double* Lr = Lm + (r0 + r) * LD;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
                for (int k = c0; k < r0; k += 2) {
                    const double2 lv = *reinterpret_cast<const double2*>(Lr + k);
                    s0 = fma(lv.x, Wm[k * LD + c0 + c], s0);
                    s1 = fma(lv.y, Wm[(k + 1) * LD + c0 + c], s1);
                }
                Pq[r * NB + c] = s0 + s1;
                wave_lds_fence();
                const double* T = Tb + I * NB * NB + r * NB;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k) acc = fma(T[k], Pq[k * NB + c], acc);
                wave_lds_fence();
                Wm[(r0 + r) * LD + c0 + c] = -acc;
                wave_lds_fence();
            }
        }
    }
    __syncthreads();

    VGPA_STAMP(6);
    // ---- phase 6 ---------------------------------------------------------------
    if (tid < D) {
        const int l = tid;
        double y0 = 0.0, y1 = 0.0;
#pragma unroll 4
        for (int k = 0; k < D; k += 2) {
            y0 = fma(Wm[k * LD + l], uv[k], y0);
            y1 = fma(Wm[(k + 1) * LD + l], uv[k + 1], y1);
        }
        a.dEdm[e * D + l] = 0.5 * w1 * cs * (y0 + y1);
    }
#pragma unroll
    for (int q0 = 0; q0 < DD / 2; q0 += NT) {
        const int q = q0 + tid;
        if (q < DD / 2) {
            const int k = (2 * q) / D, o = k * LD + ((2 * q) % D);
            const double gk = gv[k];
            const double2 w = *reinterpret_cast<const double2*>(Wm + o);
            *reinterpret_cast<double2*>(Lm + o) = make_double2(gk * w.x, gk * w.y);
        }
    }
    __syncthreads();

    VGPA_STAMP(7);
    // ---- phase 7: dEsde/dS = W' V, upper tiles, mirrored -----------------------
    if (tid < C::NUPPER) {
        // tile number -> (ti, tj): column tile tj starts at tile tj (tj + 1); long k-loops
        // (small tj) come first so the two active wavefronts see similar trip counts
        // within themselves.
        int tj = 0;
#pragma unroll
        for (int q = 1; q < NTJ; ++q) tj += (tid >= q * (q + 1)) ? 1 : 0;
        const int ti = tid - tj * (tj + 1);
        const int i0 = ti * TI, j0 = tj * TJ;
        {
            double acc[TI][TJ] = {};
            const int k0 = (i0 > j0) ? i0 : j0;
#pragma unroll 4
            for (int k = k0; k < D; ++k) {
                const double2 wv = *reinterpret_cast<const double2*>(Wm + k * LD + i0);
                const double2 v0 = *reinterpret_cast<const double2*>(Lm + k * LD + j0);
                const double2 v1 = *reinterpret_cast<const double2*>(Lm + k * LD + j0 + 2);
                const double vs[TJ] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
                for (int c = 0; c < TJ; ++c) {
                    acc[0][c] = fma(wv.x, vs[c], acc[0][c]);
                    acc[1][c] = fma(wv.y, vs[c], acc[1][c]);
                }
            }
            double* Gd = a.dEdS + e * DD;
            if (i0 + TI - 1 <= j0) {
                // tile strictly above the diagonal: full rows, and its mirror image
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    double2* row = reinterpret_cast<double2*>(Gd + (i0 + r) * D + j0);
                    row[0] = make_double2(acc[r][0], acc[r][1]);
                    row[1] = make_double2(acc[r][2], acc[r][3]);
                }
#pragma unroll
                for (int c = 0; c < TJ; ++c)
                    *reinterpret_cast<double2*>(Gd + (j0 + c) * D + i0) = make_double2(acc[0][c], acc[1][c]);
            } else {
#pragma unroll
                for (int r = 0; r < TI; ++r)
#pragma unroll
                    for (int c = 0; c < TJ; ++c) {
                        const int i = i0 + r, j = j0 + c;
                        if (i <= j) {
                            Gd[i * D + j] = acc[r][c];
                            if (i < j) Gd[j * D + i] = acc[r][c];
                        }
                    }
            }
        }
    }
    VGPA_STAMP(8);
}
