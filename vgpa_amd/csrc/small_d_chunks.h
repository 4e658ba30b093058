// D = 2, 3 scans parallel in time, for batches too small to fill the chip with one problem per
// group of four lanes (the Lorenz-63 grid cut over eight GPUs: 512 problems per GPU).
//
// A step of every integrator of core/fwd_ode.py / core/bwd_ode.py maps the state (m, S) -- or
// (lam, Psi) -- of one row to the next by an AFFINE function of the state, and m and S do not mix.
// The time axis of a problem is cut into C = 16 chunks.  One workgroup per problem, sixteen lanes
// per chunk:
//
//   phase 1  lane j < D*D of a chunk pushes the basis state (m = e_j, S = E_j) through the
//            chunk's steps with the forcing terms (b, Sigma; dEsde/dm, dEsde/dS, the jumps) set to
//            zero, lane D*D the zero state with the forcing: together the chunk's affine map,
//            obtained with the ordinary step function (no operator algebra).
//   phase 2  one wavefront walks the sixteen chunks and applies the maps to the initial state:
//            the state at the start of every chunk.
//   phase 3  every chunk is integrated again from its start state with the ordinary step
//            function and the rows are stored (one lane per chunk).
//
// A row therefore differs from a row-by-row run only through the start state of its chunk
// (rounding in another order, a few 1e-16 relative per chunk).  Two passes of ceil((N - 1) / 16)
// dependent steps replace one pass of N - 1.  Same step arithmetic as the other small-D kernels
// (SmallD::fwd_step / bwd_step).
//
// SYM (the caller's promise that the state is exactly symmetric: S0, Sigma forward; dEsde/dS, the jumps and
// the terminal state backward): a basis of symmetric matrices has D (D + 1) / 2 members, so a chunk needs
// eight lanes instead of sixteen and a workgroup cuts the time axis into 32 chunks.  The step function is the
// general one -- it maps an exactly symmetric state to an exactly symmetric state -- so the rows are those of the
// sixteen-chunk kernel up to the chunk start states.
//
// Input rows reach the lanes through LDS: the sixteen lanes of a chunk copy the rows of the next T steps
// (two buffers per chunk) while the current ones are integrated, so the latency of a global load
// (about 2000 cycles here, as long as a step) is paid once per T steps and hidden behind them.  Read
// straight into registers a few rows ahead, every step ended up waiting for the memory system.
//
// Included by scan.hip after small_d.h, inside its anonymous namespace.

template <bool SYM>
struct SmallChunksT {
    static constexpr int NT = 256, WAVES = NT / 64;
    static constexpr int ROW = SYM ? 8 : 16;          // lanes per chunk
    static constexpr int C = NT / ROW, CPW = 64 / ROW;   // chunks per workgroup / per wavefront
    static constexpr int T_FWD = SYM ? 4 : 8, T_BWD = SYM ? 2 : 4;       // steps per staged tile (LDS: two workgroups per CU)
    // members of the basis the state is expanded in (the forcing role comes after them)
    static constexpr int nbasis(int D) { return SYM ? D * (D + 1) / 2 : D * D; }
    // member q: the matrix with ones at (i, j) and (j, i); general basis: q = i D + j
    template <int D>
    static __device__ __forceinline__ void member(int q, int& i, int& j) {
        if (SYM) {
            i = 0;
            int left = q;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const bool next = left >= D - k && i == k;
                left -= next ? D - k : 0;
                i += next ? 1 : 0;
            }
            j = i + left;
        } else {
            i = q / D;
            j = q % D;
        }
    }
};
struct SmallChunks {
    // worth it when the sequential kernels leave most of the chip idle and a chunk has some length
    static bool use(int P, int N) { return P <= 1024 && N - 1 >= 4 * 32; }
};

// One chunk's staging area: ROWS rows of RW doubles, copied by the chunk's LPC lanes.
template <int RW, int ROWS, int LPC>
struct ChunkTile {
    static constexpr int TILE = RW * ROWS, NL = (TILE + LPC - 1) / LPC;
    // element e = r + LPC j of the tile: (row e / RW, entry e % RW); `src(row_index, entry)` names its address
    template <typename Src>
    static __device__ __forceinline__ void fetch(int r, Src src, double (&pre)[NL]) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = min(r + LPC * j, TILE - 1);
            pre[j] = *src(e / RW, e % RW);
        }
    }
    static __device__ __forceinline__ void commit(int r, const double (&pre)[NL], double* tile) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = r + LPC * j;
            if (j < NL - 1 || e < TILE) tile[e] = pre[j];
        }
    }
};

__device__ __forceinline__ void chunk_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <int D, int METHOD, bool SYM>
__global__ void __launch_bounds__(256, 2)
fwd_scan_small_chunks_kernel(int N, const double* __restrict__ A, const double* __restrict__ b,
                             const double* __restrict__ m0, const double* __restrict__ S0,
                             const double* __restrict__ Sigma, int sigma_stride, double dt,
                             double* __restrict__ m_out, double* __restrict__ S_out) {
    using T = SmallD<D>;
    using K = SmallChunksT<SYM>;
    constexpr int DD = T::DD, NS = D + DD, TS = K::T_FWD, RW = DD + D, NB = K::nbasis(D);
    using Tile = ChunkTile<RW, TS + 1, K::ROW>;                    // rows t .. t + TS of [A | b]
    __shared__ double ops[K::C][NB + 1][NS];     // what the roles' states became over the chunk
    __shared__ double starts[K::C][NS];          // the state at the first row of every chunk
    __shared__ double tiles[K::WAVES][2][K::CPW][Tile::TILE];
    const int tid = threadIdx.x, c = tid / K::ROW, r = tid % K::ROW;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    const double* bp = b + (size_t)p * N * D;
    double* mo = m_out + (size_t)p * N * D;
    double* So = S_out + (size_t)p * N * DD;
    const int steps = N - 1, Lc = (steps + K::C - 1) / K::C;
    const int s0 = min(c * Lc, steps), s1 = min(s0 + Lc, steps);     // this chunk: rows s0 -> s1
    double* tile0 = &tiles[tid >> 6][0][c & (K::CPW - 1)][0];
    double* tile1 = &tiles[tid >> 6][1][c & (K::CPW - 1)][0];
    double Sig[DD];
    T::load(Sigma + (size_t)p * sigma_stride, Sig);

    // rows s0 .. s1 with their forcing scaled by `aff`; store: write the rows
    auto integrate = [&](double (&S)[DD], double (&m)[D], double aff, bool store) __attribute__((always_inline)) {
        double Sg[DD], pre[Tile::NL];
#pragma unroll
        for (int q = 0; q < DD; ++q) Sg[q] = aff * Sig[q];
        const int ntile = (s1 - s0 + TS - 1) / TS;
        auto fetch = [&](int k) __attribute__((always_inline)) {
            const int first = s0 + TS * k;
            Tile::fetch(r, [&](int i, int q) {
                const int row = min(first + i, N - 1);
                return q < DD ? Ap + (size_t)row * DD + q : bp + (size_t)row * D + (q - DD);
            }, pre);
        };
        if (ntile > 0) {
            fetch(0);
            Tile::commit(r, pre, tile0);
            chunk_fence();
            if (ntile > 1) fetch(1);
        }
        for (int k = 0; k < ntile; ++k) {
            const double* cur = (k & 1) ? tile1 : tile0;
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                const int t = s0 + TS * k + i;
                if (t < s1) {
                    double At[DD], An[DD], bt[D], bn[D];
                    T::load(cur + i * RW, At);
                    T::load(cur + (i + 1) * RW, An);
#pragma unroll
                    for (int q = 0; q < D; ++q) { bt[q] = aff * cur[i * RW + DD + q]; bn[q] = aff * cur[(i + 1) * RW + DD + q]; }
                    T::template fwd_step<METHOD>(At, bt, An, bn, Sg, dt, S, m);
                    if (store) {
                        T::store(So + (size_t)(t + 1) * DD, S);
                        T::store(mo + (size_t)(t + 1) * D, m);
                    }
                }
            }
            if (k + 1 < ntile) {
                // the other buffer was last read two tiles ago (program order within the wavefront)
                Tile::commit(r, pre, (k & 1) ? tile0 : tile1);
                chunk_fence();
                if (k + 2 < ntile) fetch(k + 2);
            }
        }
    };

    // ---- phase 1: the chunk's affine map --------------------------------------------------------
    {
        const int role = min(r, NB);
        int bi, bj;
        K::template member<D>(min(role, NB - 1), bi, bj);
        double S[DD], m[D];
#pragma unroll
        for (int q = 0; q < DD; ++q)
            S[q] = (role < NB && (q == bi * D + bj || (SYM && q == bj * D + bi))) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) m[q] = (q == role) ? 1.0 : 0.0;
        integrate(S, m, role == NB ? 1.0 : 0.0, false);
        if (r <= NB) {
#pragma unroll
            for (int q = 0; q < D; ++q) ops[c][r][q] = m[q];
#pragma unroll
            for (int q = 0; q < DD; ++q) ops[c][r][D + q] = S[q];
        }
    }
    __syncthreads();
    // ---- phase 2: start states, chunk after chunk (entry e of the state per lane) --------------------
    if (tid < 64) {
        const int e = min(tid, NS - 1);
        double x = e < D ? m0[(size_t)p * D + e] : S0[(size_t)p * DD + (e - D)];
        if (tid < D) mo[tid] = x;
        else if (tid < NS) So[tid - D] = x;
        for (int k = 0; k < K::C; ++k) {
            if (tid < NS) starts[k][e] = x;
            chunk_fence();
            double acc = ops[k][NB][e];
            if (e < D) {
#pragma unroll
                for (int j = 0; j < D; ++j) acc = fma(starts[k][j], ops[k][j][e], acc);
            } else {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    int bi, bj;
                    K::template member<D>(q, bi, bj);
                    acc = fma(starts[k][D + bi * D + bj], ops[k][q][e], acc);
                }
            }
            x = acc;
            chunk_fence();
        }
    }
    __syncthreads();
    // ---- phase 3: the rows, from the chunk's start state -----------------------------------------
    {
        double S[DD], m[D];
#pragma unroll
        for (int q = 0; q < D; ++q) m[q] = starts[c][q];
#pragma unroll
        for (int q = 0; q < DD; ++q) S[q] = starts[c][D + q];
        integrate(S, m, 1.0, r == 0);
    }
}

template <int D, int METHOD, bool SYM>
__global__ void __launch_bounds__(256, 2)
bwd_scan_small_chunks_kernel(int N, const double* __restrict__ A, const double* __restrict__ dEdm,
                             const double* __restrict__ dEdS, const double* __restrict__ jm,
                             const int* __restrict__ obs_flag, const double* __restrict__ jumpS, int jumpS_stride,
                             double dt, const double* __restrict__ lam_last, const double* __restrict__ Psi_last,
                             double* __restrict__ lam_out, double* __restrict__ Psi_out) {
    using T = SmallD<D>;
    using K = SmallChunksT<SYM>;
    constexpr int DD = T::DD, NS = D + DD, TS = K::T_BWD, NB = K::nbasis(D);
    // a staged row: A, dEsde/dS, dEsde/dm, the jump of lam, the observation flag (as 0.0 / 1.0)
    constexpr int OG = DD, Og = 2 * DD, OJ = 2 * DD + D, OF = 2 * DD + 2 * D, RW = OF + 1;
    using Tile = ChunkTile<RW, TS + 1, K::ROW>;            // rows t, t - 1, .. t - TS
    __shared__ double ops[K::C][NB + 1][NS];
    __shared__ double starts[K::C][NS];
    __shared__ double tiles[K::WAVES][2][K::CPW][Tile::TILE];
    const int tid = threadIdx.x, c = tid / K::ROW, r = tid % K::ROW;
    const int p = blockIdx.x;
    const double* Ap = A + (size_t)p * N * DD;
    const double* gm = dEdm + (size_t)p * N * D;
    const double* gS = dEdS + (size_t)p * N * DD;
    const double* jmp = jm + (size_t)p * N * D;
    double* lo = lam_out + (size_t)p * N * D;
    double* Po = Psi_out + (size_t)p * N * DD;
    const bool flagged = jumpS && obs_flag;
    // step s of the scan goes from row N - 1 - s to row N - 2 - s; this chunk takes steps s0 .. s1 - 1
    const int steps = N - 1, Lc = (steps + K::C - 1) / K::C;
    const int s0 = min(c * Lc, steps), s1 = min(s0 + Lc, steps);
    double* tile0 = &tiles[tid >> 6][0][c & (K::CPW - 1)][0];
    double* tile1 = &tiles[tid >> 6][1][c & (K::CPW - 1)][0];
    double jt[DD];
#pragma unroll
    for (int q = 0; q < DD; ++q) jt[q] = flagged ? jumpS[(size_t)p * jumpS_stride + q] : 0.0;
    // the flag array as doubles would need a conversion per element: flags are fetched apart (lane i < TS + 1)
    auto integrate = [&](double (&X)[DD], double (&x)[D], double aff, bool store) __attribute__((always_inline)) {
        double pre[Tile::NL];
        const int ntile = (s1 - s0 + TS - 1) / TS;
        auto fetch = [&](int k) __attribute__((always_inline)) {
            const int first = N - 1 - (s0 + TS * k);               // row of slot 0; slot i holds row first - i
            Tile::fetch(r, [&](int i, int q) {
                const int row = max(first - i, 0);
                // (the flag slot reads a valid double; it is overwritten at commit time)
                return q < OG ? Ap + (size_t)row * DD + q
                     : q < Og ? gS + (size_t)row * DD + (q - OG)
                     : q < OJ ? gm + (size_t)row * D + (q - Og)
                     : q < OF ? jmp + (size_t)row * D + (q - OJ) : jmp + (size_t)row * D;
            }, pre);
        };
        int fl = 0;
        auto fetch_flag = [&](int k) __attribute__((always_inline)) {
            const int row = max(N - 1 - (s0 + TS * k) - min(r, TS), 0);
            fl = flagged ? obs_flag[row] : 0;
        };
        auto commit = [&](double* tile) __attribute__((always_inline)) {
            Tile::commit(r, pre, tile);
            chunk_fence();                                       // the flag slot is written after the copy
            if (r <= TS) tile[r * RW + OF] = fl ? 1.0 : 0.0;
            chunk_fence();
        };
        if (ntile > 0) {
            fetch(0);
            fetch_flag(0);
            commit(tile0);
            if (ntile > 1) { fetch(1); fetch_flag(1); }
        }
        for (int k = 0; k < ntile; ++k) {
            const double* cur = (k & 1) ? tile1 : tile0;
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                const int s = s0 + TS * k + i;
                if (s < s1) {
                    const double* rt = cur + i * RW;               // row t
                    const double* rp = cur + (i + 1) * RW;         // row t - 1, the row the step arrives at
                    double At[DD], Ap1[DD], Gt[DD], Gp[DD], gt[D], gp[D];
                    T::load(rt, At);
                    T::load(rp, Ap1);
#pragma unroll
                    for (int q = 0; q < DD; ++q) { Gt[q] = aff * rt[OG + q]; Gp[q] = aff * rp[OG + q]; }
#pragma unroll
                    for (int q = 0; q < D; ++q) { gt[q] = aff * rt[Og + q]; gp[q] = aff * rp[Og + q]; }
                    T::template bwd_step<METHOD>(At, Gt, gt, Ap1, Gp, gp, dt, X, x);
                    // jumps of the row arrived at (core/bwd_ode.py:124, :127)
                    const double on = aff * rp[OF];
#pragma unroll
                    for (int q = 0; q < DD; ++q) X[q] += on * jt[q];
#pragma unroll
                    for (int q = 0; q < D; ++q) x[q] += aff * rp[OJ + q];
                    if (store) {
                        T::store(Po + (size_t)(N - 2 - s) * DD, X);
                        T::store(lo + (size_t)(N - 2 - s) * D, x);
                    }
                }
            }
            if (k + 1 < ntile) {
                commit((k & 1) ? tile0 : tile1);
                if (k + 2 < ntile) { fetch(k + 2); fetch_flag(k + 2); }
            }
        }
    };

    {
        const int role = min(r, NB);
        int bi, bj;
        K::template member<D>(min(role, NB - 1), bi, bj);
        double X[DD], x[D];
#pragma unroll
        for (int q = 0; q < DD; ++q)
            X[q] = (role < NB && (q == bi * D + bj || (SYM && q == bj * D + bi))) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) x[q] = (q == role) ? 1.0 : 0.0;
        integrate(X, x, role == NB ? 1.0 : 0.0, false);
        if (r <= NB) {
#pragma unroll
            for (int q = 0; q < D; ++q) ops[c][r][q] = x[q];
#pragma unroll
            for (int q = 0; q < DD; ++q) ops[c][r][D + q] = X[q];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int e = min(tid, NS - 1);
        double x = e < D ? (lam_last ? lam_last[(size_t)p * D + e] : 0.0)
                         : (Psi_last ? Psi_last[(size_t)p * DD + (e - D)] : 0.0);
        if (tid < D) lo[(size_t)(N - 1) * D + tid] = x;
        else if (tid < NS) Po[(size_t)(N - 1) * DD + (tid - D)] = x;
        for (int k = 0; k < K::C; ++k) {
            if (tid < NS) starts[k][e] = x;
            chunk_fence();
            double acc = ops[k][NB][e];
            if (e < D) {
#pragma unroll
                for (int j = 0; j < D; ++j) acc = fma(starts[k][j], ops[k][j][e], acc);
            } else {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    int bi, bj;
                    K::template member<D>(q, bi, bj);
                    acc = fma(starts[k][D + bi * D + bj], ops[k][q][e], acc);
                }
            }
            x = acc;
            chunk_fence();
        }
    }
    __syncthreads();
    {
        double X[DD], x[D];
#pragma unroll
        for (int q = 0; q < D; ++q) x[q] = starts[c][q];
#pragma unroll
        for (int q = 0; q < DD; ++q) X[q] = starts[c][D + q];
        integrate(X, x, 1.0, r == 0);
    }
}
