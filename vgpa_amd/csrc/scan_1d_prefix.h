// D = 1 scans as parallel prefix computations, one workgroup per problem.
//
// In one dimension every integrator of core/fwd_ode.py (euler_1D :135, heun_1D :247, rk2_1D :367,
// rk4_1D :493) and core/bwd_ode.py (euler_1D :132, heun_1D :241, rk2_1D :367, rk4_1D :497) maps the
// state of one row to the next by an AFFINE function x -> a x + c whose coefficients depend on the
// inputs of the two rows only.  Affine maps compose associatively, so the N - 1 steps of a problem
// need not be taken one after the other:
//
//   a tile is NT * L consecutive steps; thread i owns L of them.
//   pass 1   each thread composes the maps of its steps (the coefficients of a step are taken
//            from the step function itself: c = step(0), a = step(1) - c);
//   scan     a workgroup-wide prefix composition gives every thread the state at the start of
//            its segment;
//   pass 2   each thread takes its L steps from that state with the ordinary step function and
//            stores the rows.
//
// Within a segment the arithmetic is the sequential one; the segment start states differ from a
// sequential run by rounding in a different order (a few 1e-16 relative per composition; the
// parity tests hold the results to 1e-9 against the reference's values).  All global accesses
// are coalesced through LDS (the lane-per-problem kernels read 64 cache lines per instruction),
// the next tile is in flight in registers during the current one, and a single problem uses 256
// lanes instead of one.
//
// Included by scan.hip inside its anonymous namespace.

struct Prefix1D {
    static constexpr int NT = 256, L = 8, TILE = NT * L;
    static constexpr int ARR = TILE + TILE / L + 2 + L;          // padded: element e sits at e + e / L
    static constexpr int SCRATCH = 4 * (NT / 64) + 2;            // wavefront totals of the scan, carried state
    static constexpr size_t fwd_lds_bytes() { return sizeof(double) * (2 * ARR + SCRATCH); }
    static constexpr size_t bwd_lds_bytes() { return sizeof(double) * (4 * ARR + SCRATCH); }
    static __device__ __forceinline__ int pad(int e) { return e + e / L; }
};

struct Affine2 {             // two independent maps: the mean-like and the variance-like sequence
    double a0, c0, a1, c1;
};
// later o earlier
__device__ __forceinline__ Affine2 compose(const Affine2& later, const Affine2& earlier) {
    return {later.a0 * earlier.a0, fma(later.a0, earlier.c0, later.c0), later.a1 * earlier.a1,
            fma(later.a1, earlier.c1, later.c1)};
}
__device__ __forceinline__ Affine2 shfl_up(const Affine2& v, int d) {
    return {__shfl_up(v.a0, d, 64), __shfl_up(v.c0, d, 64), __shfl_up(v.a1, d, 64), __shfl_up(v.c1, d, 64)};
}

// Exclusive prefix composition over the NT threads of the workgroup (thread order = time order).
// `tot` is LDS scratch of 4 * (NT / 64) doubles.  Returns the map from the tile's start state to the
// start state of this thread's segment.
__device__ __forceinline__ Affine2 block_exclusive(Affine2 v, double* tot, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Affine2 o = shfl_up(v, d);
        if (lane >= d) v = compose(v, o);
    }
    if (lane == 63) {
        tot[4 * w] = v.a0; tot[4 * w + 1] = v.c0; tot[4 * w + 2] = v.a1; tot[4 * w + 3] = v.c1;
    }
    Affine2 ex = shfl_up(v, 1);
    if (lane == 0) ex = {1.0, 0.0, 1.0, 0.0};
    __syncthreads();
    for (int q = w - 1; q >= 0; --q) {
        // wavefronts 0 .. w - 1 come before: ex o (tot[w-1] o ... o tot[0]), applied right to left
        const Affine2 t = {tot[4 * q], tot[4 * q + 1], tot[4 * q + 2], tot[4 * q + 3]};
        ex = compose(ex, t);
    }
    return ex;
}

// core/fwd_ode.py :135-177 (Euler), :247-307 (Heun), :367-422 (RK2), :493-559 (RK4), one step t -> t + 1
template <int METHOD>
__device__ __forceinline__ void fwd_step_1d(double At, double bt, double An, double bn, double Sig, double dt,
                                            double& m, double& S) {
    const double h = 0.5 * dt;
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
}

// core/bwd_ode.py :132-172 (Euler), :241-301 (Heun), :367-423 (RK2), :497-563 (RK4), one step t -> t - 1
// without the jumps (t: row t, p: row t - 1)
template <int METHOD>
__device__ __forceinline__ void bwd_step_1d(double At, double gmt, double gSt, double Ap, double gmp, double gSp,
                                            double dt, double& lam, double& Psi) {
    const double h = 0.5 * dt;
    double ln, Pn;
    if (METHOD == VGPA_EULER) {
        ln = lam - (-gmt + At * lam) * dt;
        Pn = Psi - (-gSt + 2.0 * Psi * At) * dt;
    } else if (METHOD == VGPA_HEUN) {
        double ftp = -gmt + At * lam;
        double ftc = -gmp + Ap * (lam - ftp * dt);
        ln = lam - h * (ftp + ftc);
        ftp = -gSt + 2.0 * Psi * At;
        ftc = -gSp + 2.0 * (Psi - ftp * dt) * Ap;
        Pn = Psi - h * (ftp + ftc);
    } else if (METHOD == VGPA_RK2) {
        const double Ak = 0.5 * (Ap + At), gmk = 0.5 * (gmp + gmt), gSk = 0.5 * (gSp + gSt);
        const double lamk = lam - h * (-gmt + At * lam);
        ln = lam - (-gmk + Ak * lamk) * dt;
        const double Psik = Psi - h * (-gSt + 2.0 * Psi * At);
        Pn = Psi - (-gSk + 2.0 * Psik * Ak) * dt;
    } else {
        const double Ak = 0.5 * (Ap + At), gmk = 0.5 * (gmp + gmt), gSk = 0.5 * (gSp + gSt);
        const double K1 = (-gmt + At * lam) * dt;
        const double K2 = (-gmk + Ak * (lam - 0.5 * K1)) * dt;
        const double K3 = (-gmk + Ak * (lam - 0.5 * K2)) * dt;
        const double K4 = (-gmp + Ap * (lam - K3)) * dt;
        ln = lam - (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
        const double L1 = (-gSt + 2.0 * Psi * At) * dt;
        const double L2 = (-gSk + 2.0 * (Psi - 0.5 * L1) * Ak) * dt;
        const double L3 = (-gSk + 2.0 * (Psi - 0.5 * L2) * Ak) * dt;
        const double L4 = (-gSp + 2.0 * (Psi - L3) * Ap) * dt;
        Pn = Psi - (L1 + 2.0 * (L2 + L3) + L4) / 6.0;
    }
    lam = ln;
    Psi = Pn;
}

template <int METHOD>
__global__ void __launch_bounds__(Prefix1D::NT)
fwd_scan_1d_prefix_kernel(int N, const double* __restrict__ A, const double* __restrict__ b,
                          const double* __restrict__ m0, const double* __restrict__ S0,
                          const double* __restrict__ Sigma, int sigma_stride, double dt,
                          double* __restrict__ m_out, double* __restrict__ S_out) {
    using C = Prefix1D;
    constexpr int NT = C::NT, L = C::L, TILE = C::TILE;
    extern __shared__ double lds[];
    double* sA = lds;
    double* sb = sA + C::ARR;
    double* tot = sb + C::ARR;
    double* carry = tot + 4 * (NT / 64);
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N;
    const double* bp = b + (size_t)p * N;
    double* mo = m_out + (size_t)p * N;
    double* So = S_out + (size_t)p * N;
    const double Sig = Sigma[(size_t)p * sigma_stride];
    double m = m0[p], S = S0[p];                      // state at the start of the tile
    if (tid == 0) {
        mo[0] = m;
        So[0] = S;
    }
    const int nsteps = N - 1;
    // rows t0 .. t0 + TILE (one more than steps) of A and b: thread j moves rows j, j + NT, ...
    double pa[L + 1], pb[L + 1];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const int e = tid + NT * k;
            const int r = min(t0 + e, N - 1);
            if (k < L || tid == 0) {
                pa[k] = Ap[r];
                pb[k] = bp[r];
            }
        }
    };
    fetch(0);
    for (int t0 = 0; t0 < nsteps; t0 += TILE) {
        __syncthreads();                              // the previous tile's rows have left LDS
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const int e = tid + NT * k;
            if (k < L || tid == 0) {
                sA[C::pad(e)] = pa[k];
                sb[C::pad(e)] = pb[k];
            }
        }
        __syncthreads();
        double a[L + 1], c[L + 1];
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            a[k] = sA[C::pad(tid * L + k)];
            c[k] = sb[C::pad(tid * L + k)];
        }
        if (t0 + TILE < nsteps) fetch(t0 + TILE);
        const int nvalid = min(L, nsteps - (t0 + tid * L));      // steps of this thread in range (may be <= 0)
        // pass 1
        Affine2 v = {1.0, 0.0, 1.0, 0.0};
#pragma unroll
        for (int k = 0; k < L; ++k) {
            if (k < nvalid) {
                double x0 = 0.0, y0 = 0.0, x1 = 1.0, y1 = 1.0;
                fwd_step_1d<METHOD>(a[k], c[k], a[k + 1], c[k + 1], Sig, dt, x0, y0);
                fwd_step_1d<METHOD>(a[k], c[k], a[k + 1], c[k + 1], Sig, dt, x1, y1);
                v = compose(Affine2{x1 - x0, x0, y1 - y0, y0}, v);
            }
        }
        const Affine2 ex = block_exclusive(v, tot, tid);
        double ms = fma(ex.a0, m, ex.c0), Ss = fma(ex.a1, S, ex.c1);
        __syncthreads();                              // every thread has its rows in registers and has read tot
        // pass 2: rows t0 + tid L + k + 1
#pragma unroll
        for (int k = 0; k < L; ++k) {
            if (k < nvalid) {
                fwd_step_1d<METHOD>(a[k], c[k], a[k + 1], c[k + 1], Sig, dt, ms, Ss);
                sA[C::pad(tid * L + k)] = ms;
                sb[C::pad(tid * L + k)] = Ss;
            }
        }
        if (nvalid >= 1 && t0 + tid * L + nvalid == min(nsteps, t0 + TILE)) {
            // the thread that took the last step of the tile
            carry[0] = ms;
            carry[1] = Ss;
        }
        __syncthreads();
        m = carry[0];
        S = carry[1];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int e = tid + NT * k;
            if (t0 + e < nsteps) {
                mo[t0 + e + 1] = sA[C::pad(e)];
                So[t0 + e + 1] = sb[C::pad(e)];
            }
        }
    }
}

template <int METHOD>
__global__ void __launch_bounds__(Prefix1D::NT)
bwd_scan_1d_prefix_kernel(int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                          const double* __restrict__ dEdS, const double* __restrict__ jm,
                          const double* __restrict__ jS_dense, const int* __restrict__ obs_flag,
                          const double* __restrict__ jumpS, int jumpS_stride, double dt,
                          const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                          double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    using C = Prefix1D;
    constexpr int NT = C::NT, L = C::L, TILE = C::TILE;
    extern __shared__ double lds[];
    double* sA = lds;
    double* sg = sA + C::ARR;
    double* sG = sg + C::ARR;
    double* sj = sG + C::ARR;
    double* tot = sj + C::ARR;
    double* carry = tot + 4 * (NT / 64);
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N;
    const double* gm = dEdm + (size_t)p * N;
    const double* gS = dEdS + (size_t)p * N;
    const double* jmp = jm + (size_t)p * N;
    const double* jSd = jS_dense ? jS_dense + (size_t)p * N : nullptr;
    const double jval = (!jSd && jumpS && obs_flag) ? jumpS[(size_t)p * jumpS_stride] : 0.0;
    double* lo = lam_out + (size_t)p * N;
    double* Po = Psi_out + (size_t)p * N;
    double lam = lam_last ? lam_last[p] : 0.0, Psi = Psi_last ? Psi_last[p] : 0.0;
    if (tid == 0) {
        lo[N - 1] = lam;
        Po[N - 1] = Psi;
    }
    const int nsteps = N - 1;
    // Step s of the scan goes from row N - 1 - s to row N - 2 - s.  Element e of a tile that starts at
    // step s0 is row N - 1 - s0 - e (TILE + 1 rows); the jump added after step s belongs to row N - 2 - s.
    double pa[L + 1], pg[L + 1], pG[L + 1], pj[L];
    auto fetch = [&](int s0) {
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const int e = tid + NT * k;
            const int r = max(N - 1 - s0 - e, 0);
            if (k < L || tid == 0) {
                pa[k] = Ap[r];
                pg[k] = gm[r];
                pG[k] = gS[r];
            }
            if (k < L) {
                // jumps of the row a step arrives at: mean jump, and the variance jump where the row is observed
                const int rj = max(r - 1, 0);
                pj[k] = jmp[rj];
            }
        }
    };
    fetch(0);
    for (int s0 = 0; s0 < nsteps; s0 += TILE) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const int e = tid + NT * k;
            if (k < L || tid == 0) {
                sA[C::pad(e)] = pa[k];
                sg[C::pad(e)] = pg[k];
                sG[C::pad(e)] = pG[k];
            }
            if (k < L) sj[C::pad(e)] = pj[k];
        }
        __syncthreads();
        double a[L + 1], g[L + 1], G[L + 1], jl[L], jP[L];
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            a[k] = sA[C::pad(tid * L + k)];
            g[k] = sg[C::pad(tid * L + k)];
            G[k] = sG[C::pad(tid * L + k)];
        }
        const int nvalid = min(L, nsteps - (s0 + tid * L));
#pragma unroll
        for (int k = 0; k < L; ++k) {
            jl[k] = sj[C::pad(tid * L + k)];
            const int row = max(N - 2 - (s0 + tid * L + k), 0);           // the row step k arrives at
            jP[k] = jSd ? jSd[row] : ((obs_flag && obs_flag[row]) ? jval : 0.0);   // core/bwd_ode.py:124,127
        }
        if (s0 + TILE < nsteps) fetch(s0 + TILE);
        Affine2 v = {1.0, 0.0, 1.0, 0.0};
#pragma unroll
        for (int k = 0; k < L; ++k) {
            if (k < nvalid) {
                double x0 = 0.0, y0 = 0.0, x1 = 1.0, y1 = 1.0;
                bwd_step_1d<METHOD>(a[k], g[k], G[k], a[k + 1], g[k + 1], G[k + 1], dt, x0, y0);
                bwd_step_1d<METHOD>(a[k], g[k], G[k], a[k + 1], g[k + 1], G[k + 1], dt, x1, y1);
                v = compose(Affine2{x1 - x0, x0 + jl[k], y1 - y0, y0 + jP[k]}, v);
            }
        }
        const Affine2 ex = block_exclusive(v, tot, tid);
        double ls = fma(ex.a0, lam, ex.c0), Ps = fma(ex.a1, Psi, ex.c1);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < L; ++k) {
            if (k < nvalid) {
                bwd_step_1d<METHOD>(a[k], g[k], G[k], a[k + 1], g[k + 1], G[k + 1], dt, ls, Ps);
                ls += jl[k];
                Ps += jP[k];
                sA[C::pad(tid * L + k)] = ls;
                sg[C::pad(tid * L + k)] = Ps;
            }
        }
        if (nvalid >= 1 && s0 + tid * L + nvalid == min(nsteps, s0 + TILE)) {
            carry[0] = ls;
            carry[1] = Ps;
        }
        __syncthreads();
        lam = carry[0];
        Psi = carry[1];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int e = tid + NT * k;
            if (s0 + e < nsteps) {
                lo[N - 2 - s0 - e] = sA[C::pad(e)];
                Po[N - 2 - s0 - e] = sg[C::pad(e)];
            }
        }
    }
}
