// Lorenz-96 unscented energy kernel, compile-time dimension: one 256-thread
// workgroup per (problem, time step), all D x D operands in LDS.
//
// Phases (block barriers between them):
//   1  stage A(t), S(t), m, b into LDS (coalesced)
//   2  <f>, <df/dx>, A m - b
//   3  blocked Cholesky S = L L' (block width 8): column-block update by all
//      threads, 8x8 diagonal factor and its inverse in the registers of one
//      wavefront (cross-lane reads, no LDS), row-panel solve
//   4  G = A L with 2x4 register tiles, then the +/- sigma-point residuals of the
//      tile; per-tile partial sums of v+-_i (over d) and of <(f-g)^2>_d (over i)
//   5  reduce the partial sums; Esde(t)
//   6  W = L^-1 by block rows (reuses the 8x8 inverses of phase 3)
//   7  dEsde/dm = (w c / 2) W' (v+ - v-);  V = diag(g) W
//   8  dEsde/dS = W' V, upper tiles only, mirrored through LDS so the result is
//      bitwise symmetric, then written with coalesced stores
//
// Included by esde.hip inside its anonymous namespace.

template <int D>
struct L96Tiled {
    static constexpr int LD = D + 1;          // padded leading dimension
    static constexpr int MS = D * LD;
    static constexpr int NB = 8;              // Cholesky block width
    static constexpr int NBLK = D / NB;
    static constexpr int TI = 2, TJ = 4;      // register tile
    static constexpr int NTI = D / TI, NTJ = D / TJ;
    static constexpr int NT = 256;
    static_assert(D % NB == 0 && D % TJ == 0 && D % TI == 0, "D must be a multiple of 8");
    static_assert(NTI * NTJ <= NT, "one register tile per thread");
    static_assert(2 * NTI * D <= MS, "partial sums must fit the scratch matrix");
    // LDS doubles: Lm, Am, Xb matrices; mean-square partials; 8x8 inverses; vectors
    static constexpr int LDS_DOUBLES = 3 * MS + NTJ * D + NBLK * NB * NB + 8 * D + 8;
};

__device__ inline double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p): hardware estimate refined by two Newton steps (full double precision).
__device__ inline double rsqrt_refined(double p) {
    double r = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}

template <int D>
__global__ void __launch_bounds__(256) esde_l96_ut_tiled_kernel(EsdeArgs a) {
    using C = L96Tiled<D>;
    constexpr int LD = C::LD, MS = C::MS, NB = C::NB, NBLK = C::NBLK, NT = C::NT;
    constexpr int TI = C::TI, TJ = C::TJ, NTI = C::NTI, NTJ = C::NTJ;
    extern __shared__ double lds[];
    double* Lm = lds;                 // S -> L -> V = diag(g) W
    double* Am = lds + MS;            // A -> W = L^-1
    double* Xb = lds + 2 * MS;        // partial sums of v+-, later the symmetric output
    double* Mb = lds + 3 * MS;        // [NTJ][D] partial sums of w (r+^2 + r-^2)
    double* Tb = Mb + NTJ * D;        // [NBLK][8][8] inverses of the diagonal blocks of L
    double* mv = Tb + NBLK * NB * NB;
    double* bv = mv + D;
    double* basev = bv + D;           // A m - b
    double* sIv = basev + D;
    double* uv = sIv + D;             // v+ - v-
    double* gv = uv + D;              // (v+ + v-)/8 - Esde/2
    double* sc = gv + D;              // [0] = pivot failure flag, [1] = Esde
    const int tid = threadIdx.x;
    const size_t e = blockIdx.x;      // p * N + t
    const int p = (int)(e / a.N);
    constexpr int DD = D * D;
    const double theta = a.theta[(size_t)p * a.theta_stride];

    // ---- phase 1 ----------------------------------------------------------
    {
        const double* Ag = a.A + e * DD;
        const double* Sg = a.S + e * DD;
#pragma unroll
        for (int q0 = 0; q0 < DD; q0 += NT) {
            const int q = q0 + tid;
            if (q < DD) {
                const int o = (q / D) * LD + (q % D);
                Am[o] = Ag[q];
                Lm[o] = Sg[q];
            }
        }
        if (tid < D) {
            mv[tid] = a.m[e * D + tid];
            bv[tid] = a.b[e * D + tid];
            sIv[tid] = a.sig_inv_diag[(size_t)p * a.sig_stride + tid];
        }
        if (tid == 0) sc[0] = 0.0;
    }
    __syncthreads();

    // ---- phase 2 ----------------------------------------------------------
    if (tid < D) {
        const int l = tid;
        const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + D - 2) % D;
        double am = 0.0;
#pragma unroll 8
        for (int k = 0; k < D; ++k) am = fma(Am[l * LD + k], mv[k], am);
        const double base = am - bv[l];
        basev[l] = base;
        const double Cxx = Lm[f1 * LD + b1] - Lm[b2 * LD + b1];
        const double Ef = Cxx + (mv[f1] - mv[b2]) * mv[b1] - mv[l] + theta;
        a.Ef[e * D + l] = Ef;
        if (a.dth_t) a.dth_t[e * D + l] = Ef + base;
    }
    if (a.Edf) {
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
    __syncthreads();

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
                    s0 = fma(ri[k], rc[k], s0);
                    s1 = fma(ri[k + 1], rc[k + 1], s1);
                }
                Lm[i * LD + c0 + c] -= (s0 + s1);
            }
            __syncthreads();
        }
        // (b) factor the 8x8 diagonal block and invert its factor: wavefront 0,
        //     lane j < 8 holds column j; U = L' is built row by row.
        if (tid < 64) {
            const int j = tid & 7;
            double s[NB], rinv[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) s[i] = Lm[(c0 + j) * LD + c0 + i];   // lower triangle, as LAPACK reads it
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const double pk = readlane_f64(s[k], k);
                ok = ok && (pk > 0.0);
                rinv[k] = rsqrt_refined(pk);
                s[k] *= rinv[k];
#pragma unroll
                for (int i = k + 1; i < NB; ++i) {
                    const double uki = readlane_f64(s[k], i);
                    s[i] = fma(-uki, s[k], s[i]);
                }
            }
            // Y = U^-1 (upper), column j in this lane; T = Y' = inverse of the lower factor.
            double y[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) y[i] = 0.0;
#pragma unroll
            for (int i = NB - 1; i >= 0; --i) {
                double acc = 0.0;
#pragma unroll
                for (int k = i + 1; k < NB; ++k) acc = fma(readlane_f64(s[i], k), y[k], acc);
                y[i] = (j == i) ? rinv[i] : -rinv[i] * acc;
            }
            if (tid < NB) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (i <= j) Lm[(c0 + j) * LD + c0 + i] = s[i];     // L[j][i] = U[i][j]
                    Tb[J * NB * NB + j * NB + i] = (i <= j) ? y[i] : 0.0;   // T[j][i] = Y[i][j]
                }
                if (!ok && tid == 0) sc[0] = 1.0;
            }
        }
        __syncthreads();
        if (sc[0] != 0.0) break;   // uniform: every thread reads the same flag
        // (c) rows below the block: L[i][c0..c0+7] = S'[i][c0..c0+7] T'
        {
            const int i = c0 + NB + tid;
            if (i < D) {
                double r[NB], o[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) r[k] = Lm[i * LD + c0 + k];
                const double* T = Tb + J * NB * NB;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k <= c; ++k) acc = fma(r[k], T[c * NB + k], acc);
                    o[c] = acc;
                }
#pragma unroll
                for (int c = 0; c < NB; ++c) Lm[i * LD + c0 + c] = o[c];
            }
        }
        __syncthreads();
    }
    if (sc[0] != 0.0) {
        // Not positive definite.  The reference falls back to diagonal sigma points,
        // which needs a general inverse of S: left to the generic path's status
        // handling -- flag the step and poison its energy outputs.
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

    // ---- phase 4: G = A L and the sigma-point residuals ----------------------
    double* Vp = Xb;                  // [NTI][D]
    double* Vm = Xb + NTI * D;        // [NTI][D]
    if (tid < NTI * NTJ) {
        const int di = tid / NTJ, ij = tid % NTJ;
        const int d0 = di * TI, i0 = ij * TJ;
        double G[TI][TJ] = {};
        const double* A0 = Am + d0 * LD;
        const double* A1 = A0 + LD;
#pragma unroll 4
        for (int j = i0; j < D; ++j) {
            const double a0 = A0[j], a1 = A1[j];
            const double* Lr = Lm + j * LD + i0;
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
                const double lv = Lr[c];
                G[0][c] = fma(a0, lv, G[0][c]);
                G[1][c] = fma(a1, lv, G[1][c]);
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
        double mb[TI] = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < TJ; ++c) {
            double o[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) o[q] = cs * Lm[rr[q] * LD + i0 + c];
            double vpc = 0.0, vmc = 0.0;
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                // state index d = d0 + r sits at position q = r + 2 of the window
                const double xm2p = mm[r] + o[r], xm1p = mm[r + 1] + o[r + 1];
                const double x0p = mm[r + 2] + o[r + 2], xp1p = mm[r + 3] + o[r + 3];
                const double xm2m = mm[r] - o[r], xm1m = mm[r + 1] - o[r + 1];
                const double x0m = mm[r + 2] - o[r + 2], xp1m = mm[r + 3] - o[r + 3];
                const double fp = (xp1p - xm2p) * xm1p - x0p + theta;
                const double fm = (xp1m - xm2m) * xm1m - x0m + theta;
                const double bs = basev[d0 + r];
                const double cG = cs * G[r][c];
                const double rp = fp + bs + cG, rm = fm + bs - cG;
                const double rp2 = rp * rp, rm2 = rm * rm;
                const double si = sIv[d0 + r];
                vpc = fma(si, rp2, vpc);
                vmc = fma(si, rm2, vmc);
                mb[r] += w1 * (rp2 + rm2);
            }
            Vp[di * D + i0 + c] = vpc;
            Vm[di * D + i0 + c] = vmc;
        }
#pragma unroll
        for (int r = 0; r < TI; ++r) Mb[ij * D + d0 + r] = mb[r];
    }
    __syncthreads();

    // ---- phase 5: reductions (wavefront 0) -----------------------------------
    if (tid < 64) {
        const int l = tid;
        double vpl = 0.0, vml = 0.0, a0 = 0.0, mbar = 0.0;
        if (l < D) {
#pragma unroll
            for (int q = 0; q < NTI; ++q) {
                vpl += Vp[q * D + l];
                vml += Vm[q * D + l];
            }
            const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + D - 2) % D;
            const double f0 = (mv[f1] - mv[b2]) * mv[b1] - mv[l] + theta;
            const double r0 = f0 + basev[l];
            a0 = sIv[l] * r0 * r0;
            mbar = w0 * r0 * r0;
#pragma unroll
            for (int q = 0; q < NTJ; ++q) mbar += Mb[q * D + l];
        }
        const double v0 = wave_sum(a0);
        const double Esde = 0.5 * (w0 * v0 + w1 * wave_sum(vpl + vml));
        if (l < D) {
            uv[l] = vpl - vml;
            gv[l] = 0.125 * (vpl + vml) - 0.5 * Esde;
            if (a.dSig_t) a.dSig_t[e * D + l] = mbar;
        }
        if (l == 0) {
            a.Esde_t[e] = Esde;
            if (a.status) a.status[e] = 0;
        }
    }
    // ---- phase 6: W = L^-1 into Am, by block rows ----------------------------
    double* Wm = Am;
    __syncthreads();
#pragma unroll
    for (int q0 = 0; q0 < DD; q0 += NT) {   // zero, then the diagonal blocks
        const int q = q0 + tid;
        if (q < DD) Wm[(q / D) * LD + (q % D)] = 0.0;
    }
    __syncthreads();
    for (int q = tid; q < NBLK * NB * NB; q += NT) {
        const int J = q / (NB * NB), r = (q / NB) % NB, c = q % NB;
        Wm[(J * NB + r) * LD + J * NB + c] = Tb[q];
    }
    __syncthreads();
    double* Pb = Xb;                  // [NB][D] products sum_K L_IK W_KJ
#pragma unroll 1
    for (int I = 1; I < NBLK; ++I) {
        const int r0 = I * NB;
        const int r = tid / r0, c = tid % r0;      // r < 8 rows, c < r0 columns
        if (tid < NB * r0) {
            const double* Lr = Lm + (r0 + r) * LD;
            double s0 = 0.0, s1 = 0.0;
            // k runs over the columns of L left of the diagonal block; W[k][c] = 0 for k < c
            const int kb = (c / NB) * NB;
#pragma unroll 4
            for (int k = kb; k < r0; k += 2) {
                s0 = fma(Lr[k], Wm[k * LD + c], s0);
                s1 = fma(Lr[k + 1], Wm[(k + 1) * LD + c], s1);
            }
            Pb[r * D + c] = s0 + s1;
        }
        __syncthreads();
        if (tid < NB * r0) {
            const double* T = Tb + I * NB * NB + r * NB;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) acc = fma(T[k], Pb[k * D + c], acc);
            Wm[(r0 + r) * LD + c] = -acc;
        }
        __syncthreads();
    }

    // ---- phase 7 ---------------------------------------------------------------
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
    for (int q0 = 0; q0 < DD; q0 += NT) {
        const int q = q0 + tid;
        if (q < DD) {
            const int k = q / D, o = k * LD + (q % D);
            Lm[o] = gv[k] * Wm[o];
        }
    }
    __syncthreads();

    // ---- phase 8: dEsde/dS = W' V, upper tiles, mirrored ----------------------
    double* Ob = Xb;
    if (tid < NTI * NTJ) {
        const int ti = tid / NTJ, tj = tid % NTJ;
        const int i0 = ti * TI, j0 = tj * TJ;
        if (i0 <= j0 + TJ - 1) {
            double acc[TI][TJ] = {};
            const int k0 = (i0 > j0) ? i0 : j0;
#pragma unroll 4
            for (int k = k0; k < D; ++k) {
                const double w0k = Wm[k * LD + i0], w1k = Wm[k * LD + i0 + 1];
                const double* Vr = Lm + k * LD + j0;
#pragma unroll
                for (int c = 0; c < TJ; ++c) {
                    const double vv = Vr[c];
                    acc[0][c] = fma(w0k, vv, acc[0][c]);
                    acc[1][c] = fma(w1k, vv, acc[1][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < TI; ++r)
#pragma unroll
                for (int c = 0; c < TJ; ++c) {
                    const int i = i0 + r, j = j0 + c;
                    if (i <= j) {
                        Ob[i * LD + j] = acc[r][c];
                        Ob[j * LD + i] = acc[r][c];
                    }
                }
        }
    }
    __syncthreads();
    {
        double* Gd = a.dEdS + e * DD;
#pragma unroll
        for (int q0 = 0; q0 < DD; q0 += NT) {
            const int q = q0 + tid;
            if (q < DD) Gd[q] = Ob[(q / D) * LD + (q % D)];
        }
    }
}
