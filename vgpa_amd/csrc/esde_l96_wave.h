// Lorenz-96 unscented energy kernel for D = 40: ONE WAVEFRONT per (problem, time step),
// eight steps in flight per CU, the matrix products on the FP64 matrix cores.  Workgroups are
// pairs of wavefronts (two consecutive steps) that share nothing but the factorisation of the
// diagonal blocks: see wv_factor_diag.
//
// Everything a step needs sits in one 40 x 42 LDS matrix `buf` (its dead corners are the scratch)
// plus a few vectors, 16.3 KB in all, and in registers; the only barriers are the six around the
// three diagonal blocks.
//
//   buf, lower triangle:  c2 S(t) -> L' = chol(c2 S) -> W' = L'^-1, block by block in place
//                         (c2 = D + kappa: L' = sqrt(c2) L holds the sigma-point offsets)
//   buf, upper triangle:  the inverses T_J of the 16 x 16 diagonal blocks of L', stored
//                         transposed and shifted two columns to the right:
//                         T_J[a][b] (b <= a) is buf[16 J + b][16 J + a + 2]
//   registers:            A(t) as matrix-core operand fragments (loaded straight from
//                         global memory, in flight during the factorisation), one tile row
//                         of G' = A L' at a time, the tiles of dEsde/dS
//
// Matrix-core tiles are 16 x 16 on the 40 x 40 matrices padded to 48 x 48 (instruction
// v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
// C[(l >> 4) + 4 reg][l & 15]).  The rows of a tile are PERMUTED when they are fed in:
// operand row rho carries matrix row pi(rho) = 4 (rho & 3) + (rho >> 2), so that the four
// result registers of a lane are four CONSECUTIVE matrix rows 4 (l >> 4) + reg; the
// Lorenz-96 drift couples rows d-2 .. d+1, which a lane then reads once for all four.
// Rows / columns 40..47 do not exist: operand rows are clamped (finite duplicates) or
// read as zero and the results are never stored.
//
// Phases, all wave-synchronous:
//   1  issue the loads: m, b, Sigma^-1, S(t) (16-byte pieces, three rows per instruction,
//      to LDS times c2), A(t) fragments
//   2  blocked Cholesky, block width 16.  Diagonal block in registers: lane c of the first
//      row of 16 lanes holds column c of the block, lane c of the second row column c of
//      the accumulated row operations M; per pivot every lane eliminates its column with
//      multipliers broadcast by DPP (see wv_factor_diag), which yields U = D L~' and
//      M = L~^-1 together (S = L~ D L~'), hence L = U' D^-1/2 and T = D^-1/2 M.  Panel
//      below: L_IJ = S_IJ T_J' and trailing update S_IK -= L_IJ L_KJ' on the matrix cores.
//   3  <f>, A m - b, optionally <df/dx>
//   4  per tile row: G' = A L' on the matrix cores, then the +/- sigma-point residuals of
//      every entry in sum / difference form, reduced over the state index within the lane
//      and across the four lane groups
//   5  Esde(t), u = v+ - v-, g
//   6  W' = L'^-1 in place: W_IJ = -T_I sum_K L_IK W_KJ (matrix cores, one scratch tile);
//      the part above the diagonal is zeroed
//   7  one pass over the rows of W': dEsde/dS = W' diag(g) W on the matrix cores, W'u beside it
//      on the vector ALU (the rows are in registers anyway).
//      Both operands of the first are scaled by sqrt|g_k| (one with the sign of g_k), so
//      that entry (i, j) and entry (j, i) are sums of identical products in identical
//      order: the result is bitwise symmetric, as the symmetric backward scan requires.
//
// A step whose S(t) is not positive definite is flagged for the fallback pass exactly as
// by the workgroup-per-step kernel.  Included by esde.hip inside its anonymous namespace,
// after esde_l96_tiled.h (shares rcp_refined, rsqrt_refined, wave_lds_fence, readlane_f64).

typedef double wv_f64x4 __attribute__((ext_vector_type(4)));

#ifdef VGPA_STAMPS
#define WV_STAMP(k) do { if (blockIdx.x == 1000 && blockIdx.y == 30 && threadIdx.x == 0) g_l96_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#elif defined(VGPA_PHASE_MARKS)
#define WV_STAMP(k) asm volatile("; WVPHASE " #k)      /* a comment in the ISA listing, for instruction counts per phase */
#else
#define WV_STAMP(k) do { } while (0)
#endif

struct L96Wave {
    static constexpr int D = 40, LD = 42, MS = D * LD, DD = D * D;
    // The blocks of `buf` above the diagonal blocks are dead once S(t) is loaded (only the lower blocks
    // and the full diagonal blocks are read, and T_J shifted two columns reaches column 16 J + 17).
    // They hold the scratch:
    //   rows 0..15, columns 18..33   the 16 x 16 transposition tile of the inversion
    //   rows 0..31, columns 34..41   where stores of lanes whose row or column does not exist go (an
    //                                address select instead of an exec-mask region); lanes may share
    //                                a slot, nobody reads them
    static constexpr int X_COL = 18, DUMP_COL = 34;
    static constexpr int OFF_VEC = MS;
    static constexpr int VL = 48;                        // vectors: m, A m - b, Sigma^-1, u, sa, sb, m on the ring
    static constexpr int OFF_ZERO = OFF_VEC + 7 * VL;    // zeros: what a masked operand load reads
    static constexpr int LDS_DOUBLES = OFF_ZERO + 16;
    static constexpr size_t WG_LDS_BYTES = sizeof(double) * (2 * LDS_DOUBLES + 2);   // two steps + two flags
    // dump slots: four rows LD apart from the first / two doubles / four doubles per lane
    static __device__ __forceinline__ int dump_rows4(int lane) { return (lane >> 3) * LD + DUMP_COL + (lane & 7); }
    static __device__ __forceinline__ int dump_pair(int lane) { return ((lane & 31) >> 2) * LD + DUMP_COL + 2 * (lane & 3); }
    static __device__ __forceinline__ int dump_quad(int lane) { return (lane >> 1) * LD + DUMP_COL + 4 * (lane & 1); }
};

__device__ __forceinline__ wv_f64x4 wv_mfma(double a, double b, wv_f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Cross-lane helpers.  A DPP row rotation moves data inside each row of 16 lanes on the
// vector ALU (no LDS round trip).
template <int CTRL>
__device__ __forceinline__ double wv_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes with equal (lane >> 4), in every lane (the lanes add in different
// orders, so their copies may differ in the last bit: read ONE lane where that matters)
__device__ __forceinline__ double wv_row_sum(double v) {
    v += wv_dpp<0x128>(v);   // row_ror:8
    v += wv_dpp<0x124>(v);   // row_ror:4
    v += wv_dpp<0x122>(v);   // row_ror:2
    v += wv_dpp<0x121>(v);   // row_ror:1
    return v;
}
// sum over lanes with equal (lane & 15), in every lane: the two halves of the wavefront are
// added (v_permlane32_swap hands every lane both halves' values), then the even and odd rows of
// 16 lanes (v_permlane16_swap).  Six vector instructions; a matrix-core instruction with an
// all-ones operand does the same sum but holds the SIMD for 64 cycles.
__device__ __forceinline__ double wv_col_sum(double v) {
#ifdef VGPA_COLSUM_MFMA
    const wv_f64x4 z = {0.0, 0.0, 0.0, 0.0};
    return wv_mfma(1.0, v, z)[0];
#endif
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    return v;
}

// v_mov_b64 / v_fmac_f64 with the DPP control row_newbcast:N take their first source from
// lane N of the reading lane's own row of 16.  The two wait states a DPP read needs after
// the vector ALU wrote its source are supplied here (s_nop 1): the compiler does not look
// into inline assembly.
template <int N>
__device__ __forceinline__ double wv_row_bcast(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
    return r;
}
// acc += (t of lane N of the row) * x
template <int N, bool NOP>
__device__ __forceinline__ void wv_fmac_bcast(double& acc, double t, double x) {
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(t), "v"(x), "n"(N));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(t), "v"(x), "n"(N));
}
// rows 1 and 3 of the result are rows 0 and 2 of v; rows 0 and 2 are unchanged
__device__ __forceinline__ double wv_even_rows_to_odd(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int rlo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[0];
    const int rhi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false)[0];
    return __hiloint2double(rhi, rlo);
}

template <int K, int I, int NP>
struct WvElim {
    static __device__ __forceinline__ void run(double (&x)[NP], double ntk) {
        if constexpr (I < NP) {
            wv_fmac_bcast<I, I == K + 1>(x[I], ntk, x[K]);
            WvElim<K, I + 1, NP>::run(x, ntk);
        }
    }
};
__device__ __forceinline__ double wv_rcp1(double p) {
    double q = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, q, 1.0);
    return fma(q, e, q);
}
__device__ __forceinline__ double wv_rsqrt1(double p) {
    double r = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * r), r, 1.0);
    return fma(0.5 * r, e, r);
}
template <int K, int NP>
struct WvPivots {
    static __device__ __forceinline__ void run(double (&x)[NP], int c, double& pivc) {
        if constexpr (K < NP) {
            const double piv = wv_row_bcast<K>(x[K]);           // S rows: S[k][k]; M rows: 1
            pivc = (c == K) ? piv : pivc;
            // multipliers -S[k][i] / S[k][k] in lane i of the S rows, copied to the M rows
            const double ntk = wv_even_rows_to_odd(x[K] * -wv_rcp1(piv));
            WvElim<K, K + 1, NP>::run(x, ntk);
            WvPivots<K + 1, NP>::run(x, c, pivc);
        }
    }
};
template <int I, int NP>
struct WvScale {
    static __device__ __forceinline__ void run(double (&x)[NP], double rsc) {
        if constexpr (I < NP) {
            x[I] *= wv_row_bcast<I>(rsc);
            WvScale<I + 1, NP>::run(x, rsc);
        }
    }
};

// One 16-wide diagonal block of the factorisation (NP valid rows: 16, or 8 for the last).
// In: the block of the partly updated S in buf.  Out: L_JJ in the lower triangle of the
// block; T_J = L_JJ^-1 transposed and shifted two columns right in its upper triangle.
// Lane c of row 0 (of 16 lanes) holds column c of the block, lane c of row 1 column c of M,
// the accumulated row operations (M = I at the start); rows 2 and 3 do the same for a second block.
// Per pivot k every lane eliminates its column, x[i] -= t_i x[k], with the multipliers
// t_i = S[k][i] / S[k][k] taken by DPP broadcast from lane i of the S row.  That leaves
// U = D L~' in the S lanes and M = L~^-1 in the M lanes (S = L~ D L~'), so L = U' D^-1/2 and
// T = D^-1/2 M: both are x[i] / sqrt(d_i).  No LDS traffic and no scalar registers inside
// the pivot chain.  A pivot <= 0 is reported (a NaN pivot is not: it poisons the outputs of the
// step instead).
//
// The two halves of the wavefront factor two different blocks: lanes 0..31 the block in `buf`, lanes
// 32..63 the block at the same place `other` doubles further on in LDS (the buffer of the partner
// wavefront's step; other = 0 makes both halves factor the same block).  Bit 0 of the result is
// clear when a pivot of the first block was not positive, bit 1 for the second.
template <int J>
__device__ __forceinline__ int wv_factor_diag(double* buf0, const double* zeros0, int other, int lane) {
    constexpr int LD = L96Wave::LD, c0 = 16 * J, NP = (J == 2) ? 8 : 16;
    const int c = lane & 15;
    const bool is_m = (lane & 16) != 0;
    const int shift = (lane & 32) ? other : 0;
    double* buf = buf0 + shift;
    const double* zeros = zeros0 + shift;
    double x[NP];
    {
        // S lanes: column c = row c (symmetric); M lanes: sixteen zeros, then the one of I
        const double* src = is_m ? zeros : buf + (c0 + min(c, NP - 1)) * LD + c0;
#pragma unroll
        for (int h = 0; h < NP / 2; ++h) {
            const double2 v = *reinterpret_cast<const double2*>(src + 2 * h);
            x[2 * h] = v.x;
            x[2 * h + 1] = v.y;
        }
        const int cm = is_m ? c : -1;
#pragma unroll
        for (int i = 0; i < NP; ++i) x[i] = __hiloint2double((i == cm) ? 0x3ff00000 : __double2hiint(x[i]), __double2loint(x[i]));
    }
    double pivc = 1.0;             // pivot of row c (S rows; 1 in the M rows and the padding)
    __builtin_amdgcn_s_setprio(2);     // the pivot chain is latency-bound and leaves most issue slots to the other wavefronts
    WvPivots<0, NP>::run(x, c, pivc);
    __builtin_amdgcn_s_setprio(0);
    const double rsc = wv_even_rows_to_odd(wv_rsqrt1(pivc));
    WvScale<0, NP>::run(x, rsc);
    // S lane c: L[c][k] = x[k] for k <= c, to buf[c0 + c][c0 + k]
    // M lane c: T[i][c] = x[i] for i >= c, to buf[c0 + c][c0 + i + 2]
    // In 16-byte pieces h = 0 .. NP/2 - 1: an S lane writes the pieces h <= c / 2, an M lane the
    // pieces h >= c / 2 (the half-used piece spills into the free superdiagonal); every other
    // piece, and every lane beyond the block, goes to the lane's dump slot.
    {
        const int hc = c >> 1;
        const bool lane_writes = c < NP;                               // both halves write (equal values when other = 0)
        const unsigned first = is_m ? hc : 0;                            // pieces first .. first + count - 1
        const unsigned count = lane_writes ? (is_m ? NP / 2 - hc : hc + 1) : 0;
        double* dst = buf + (c0 + c) * LD + c0 + (is_m ? 2 : 0);
        double* nowhere = buf + L96Wave::dump_pair(lane);
#pragma unroll
        for (int h = 0; h < NP / 2; ++h) {
            const bool wr = (unsigned)h - first < count;
            *reinterpret_cast<double2*>(wr ? dst + 2 * h : nowhere) = make_double2(x[2 * h], x[2 * h + 1]);
        }
    }
    wave_lds_fence();
    // every pivot is the pivc of one S lane
    const unsigned long long bad = __builtin_amdgcn_ballot_w64(!(pivc > 0.0));
    return ((bad & 0xffffffffull) == 0 ? 1 : 0) | ((bad >> 32) == 0 ? 2 : 0);
}

// indices l + 1, l - 1, l - 2 on the ring of D = 40 components
__device__ __forceinline__ void wv_ring(int l, int& f1, int& b1, int& b2) {
    f1 = (l == L96Wave::D - 1) ? 0 : l + 1;
    b1 = (l == 0) ? L96Wave::D - 1 : l - 1;
    b2 = (l <= 1) ? l + L96Wave::D - 2 : l - 2;
}

// PACKED_S / PACKED_G: S(t) is read / dEsde/dS(t) is written in the packed symmetric layout of sym_rows.h
// (1152 doubles a step).  The kernel only ever reads the lower block triangle of S with whole
// diagonal blocks, which is what that layout holds, and forms only those tiles of dEsde/dS.
template <bool HYPER, bool PACKED_S, bool PACKED_G>
__global__ void __launch_bounds__(128, 2) esde_l96_ut_wave40_kernel(EsdeArgs a) {
    using C = L96Wave;
    constexpr int D = C::D, LD = C::LD, DD = C::DD, VL = C::VL;
    constexpr int SS = PACKED_S ? vgpa::SymRows40::SIZE : DD;    // doubles per S(t) in memory
    constexpr int SG = PACKED_G ? vgpa::SymRows40::SIZE : DD;    // doubles per dEsde/dS(t) in memory
    extern __shared__ double lds_all[];
    // Two wavefronts, two consecutive steps, each with its own LDS image.  They only meet at the three
    // diagonal blocks of the factorisation, which ONE of them factors for both (the halves of that
    // wavefront take a block each) while the other waits at the barrier.
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    double* lds = lds_all + wv * C::LDS_DOUBLES;
    const int other = wv == 0 ? C::LDS_DOUBLES : -C::LDS_DOUBLES;
    int* pd_flags = reinterpret_cast<int*>(lds_all + 2 * C::LDS_DOUBLES);     // [w]: the blocks of wavefront w were positive definite
    double* buf = lds;
    double* Xs = buf + C::X_COL;
    double* mv = lds + C::OFF_VEC;
    double* basev = mv + VL;
    double* sIv = basev + VL;
    double* uv = sIv + VL;
    double* sav = uv + VL;      // sqrt|g|
    double* sbv = sav + VL;     // sign(g) sqrt|g|
    double* mvx = sbv + VL;     // mvx[k] = m[(k - 2) mod 40], k = 0 .. 47: stencil windows need no wrap
    const double* zslot = lds + C::OFF_ZERO;       // holds 0.0: masked loads select this address
    const int lane0 = threadIdx.x & 63;
    constexpr int NQ3 = (D + 2) / 3;               // 14 load instructions of three rows of S(t) each
    const int p = blockIdx.y;                      // problem, step: no division by N
    const size_t e = (size_t)p * a.N + min(2 * (int)blockIdx.x + wv, a.N - 1);   // an odd last step is done twice
    // The lane index is made opaque again at every phase: addresses and masks derived from it
    // are then recomputed where they are used instead of being kept in registers (or spilled)
    // from one phase to the next.
    int lane = lane0, c16, g4, pi;
#define WV_RELANE()                                                                  \
    do {                                                                             \
        asm volatile("" : "+v"(lane));                                               \
        c16 = lane & 15;                                                             \
        g4 = lane >> 4;                                                              \
        pi = 4 * (c16 & 3) + (c16 >> 2); /* matrix row fed in as operand row c16 */  \
    } while (0)
    WV_RELANE();
    const double theta = a.theta[(size_t)p * a.theta_stride];

    const double kappa = 1.05 * D;
    const double c2 = D + kappa;                   // the sigma points are m +- sqrt(c2) L[:, i]
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);

    WV_STAMP(16);
    // ---- phase 1: loads; c2 S(t) goes to LDS -------------------------------------------------
    // Everything below works with L' = chol(c2 S) = sqrt(c2) L, the sigma-point offsets
    // themselves, and W' = L'^-1 = W / sqrt(c2); the factors c2 are put back at the outputs.
    // Sixteen-byte pieces: instruction i moves rows 3 i .. 3 i + 2, lane l (< 60) the piece
    // (row 3 i + l / 20, column pair l % 20); every address is one per-lane base plus a
    // compile-time offset.  Lanes 60..63 repeat lane 59.
    if constexpr (PACKED_S) {
        // nine instructions: rows 0..15 eight at a time (8 pieces a row), rows 16..31 four at a time
        // (16 pieces), rows 32..39 three at a time on 60 lanes (20 pieces without the padding); the last
        // one has rows 38, 39 only and the lanes of its third row repeat row 39
        const double2* src = reinterpret_cast<const double2*>(a.S + e * SS);
        const int l60 = min(lane, 59);
        const int lr = (l60 >= 20) + (l60 >= 40), lc = l60 - 20 * lr;
        const double2* sA = src + lane;
        const double2* sB = src + 128 + lane;
        const double2* sC = src + 384 + 24 * lr + lc;
        double* dA = buf + (lane >> 3) * LD + 2 * (lane & 7);
        double* dB = buf + (16 + (lane >> 4)) * LD + 2 * (lane & 15);
        double* dC = buf + (32 + lr) * LD + 2 * lc;
        const int back = lr == 2 ? 1 : 0;
        double2 v[9];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = sA[64 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[2 + i] = sB[64 * i];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[6 + i] = sC[72 * i];
        v[8] = sC[72 * 2 - 24 * back];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<double2*>(dA + 8 * i * LD) = make_double2(c2 * v[i].x, c2 * v[i].y);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<double2*>(dB + 4 * i * LD) = make_double2(c2 * v[2 + i].x, c2 * v[2 + i].y);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<double2*>(dC + 3 * i * LD) = make_double2(c2 * v[6 + i].x, c2 * v[6 + i].y);
        *reinterpret_cast<double2*>(dC + (6 - back) * LD) = make_double2(c2 * v[8].x, c2 * v[8].y);
    } else {
        const int l60 = min(lane, 59);
        const int lr = (l60 >= 20) + (l60 >= 40), lc = l60 - 20 * lr;
        const double2* src = reinterpret_cast<const double2*>(a.S + e * DD) + lr * (D / 2) + lc;
        double sx[NQ3], sy[NQ3];
#pragma unroll
        for (int i = 0; i < NQ3; ++i) {
            // the last instruction has row 39 only: the other lanes repeat its pieces
            const double2 v = (i < NQ3 - 1) ? src[i * 3 * (D / 2)] : src[i * 3 * (D / 2) - lr * (D / 2)];
            sx[i] = v.x;
            sy[i] = v.y;
        }
        double* dst = buf + lr * LD + 2 * lc;
#pragma unroll
        for (int i = 0; i < NQ3 - 1; ++i)
            *reinterpret_cast<double2*>(dst + i * 3 * LD) = make_double2(c2 * sx[i], c2 * sy[i]);
        *reinterpret_cast<double2*>(dst + (NQ3 - 1) * 3 * LD - lr * LD) = make_double2(c2 * sx[NQ3 - 1], c2 * sy[NQ3 - 1]);
    }
    const int lv = min(lane, D - 1);
    const double m_l = a.m[e * D + lv];
    const double b_l = a.b[e * D + lv];
    const double si_l = a.sig_inv_diag[(size_t)p * a.sig_stride + lv];
    double af[3][10];                              // A[16 i + pi][4 s + g4]
    {
        const double* Ae = a.A + e * DD;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double* row = Ae + min(16 * i + pi, D - 1) * D + g4;
#pragma unroll
            for (int s = 0; s < 10; ++s) af[i][s] = row[4 * s];
        }
    }
    if (lane < VL) {
        mvx[lane] = a.m[e * D + (lane < 2 ? lane + D - 2 : (lane < D + 2 ? lane - 2 : lane - D - 2))];
        mv[lane] = lane < D ? m_l : 0.0;
        sIv[lane] = lane < D ? si_l : 0.0;
        basev[lane] = 0.0;
    }
    if (lane < 16) lds[C::OFF_ZERO + lane] = 0.0;
    if (lane == 0) pd_flags[wv] = 1;
    wave_lds_fence();

    // the covariance term of <f>, before S is overwritten by its factor
    double Cxx = 0.0;
    {
        const int l = lv;
        int f1, b1, b2;
        wv_ring(l, f1, b1, b2);
        // packed: entries above the block diagonal are not in LDS, their mirror images are
        const int r1 = PACKED_S ? max(f1, b1) : f1, c1 = PACKED_S ? min(f1, b1) : b1;
        const int r2 = PACKED_S ? max(b2, b1) : b2, c2i = PACKED_S ? min(b2, b1) : b1;
        Cxx = (buf[r1 * LD + c1] - buf[r2 * LD + c2i]) * (1.0 / c2);
    }
    wave_lds_fence();

    WV_STAMP(17);
    // ---- phase 2: blocked Cholesky ------------------------------------------------------
    // Operand fetchers.  "rows" operands: element [r][k] of a block whose rows are matrix
    // rows; as A operand the lane feeds row pi, as B operand (the block transposed) row c16.
    auto rows4 = [&](int row, int col0, double (&o)[4]) {
        const double* src = buf + min(row, D - 1) * LD + col0 + g4;
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = src[4 * s];
    };
    // B operand T_J': element [k][c] = T_J[c][k], zero above the diagonal
    auto tjt4 = [&](int J, double (&o)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * s + g4;
            o[s] = *((k <= c16) ? buf + (16 * J + k) * LD + 16 * J + c16 + 2 : zslot);
        }
    };
    auto tile_rd = [&](int I, int K, wv_f64x4& t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = buf[min(16 * I + 4 * g4 + q, D - 1) * LD + 16 * K + c16];
    };
    // rows 16 I + 4 g4 + q, column 16 K + c16 of buf; lanes outside the 40 x 40 matrix (only
    // in block row / column 2) write to their dump slot
    auto tile_ptr = [&](int I, int K) {
        const bool valid = (I < 2 || g4 < 2) && (K < 2 || c16 < 8);
        return buf + (valid ? (16 * I + 4 * g4) * LD + 16 * K + c16 : C::dump_rows4(lane));
    };
    auto tile_wr = [&](int I, int K, const wv_f64x4& t) {
        double* dst = tile_ptr(I, K);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * LD] = t[q];
    };
    double a0 = 0.0;          // squared centre-point residual of component `lane`
    // ---- phase 3: <f>, A m - b.  Independent of the factor, so each wavefront runs it while its
    // partner factors a diagonal block for both steps (wavefront 1 during block 0, wavefront 0
    // during block 1) instead of waiting at the barrier.
    auto phase3 = [&]() __attribute__((always_inline)) {
        WV_RELANE();
        {
            double mk[10];
#pragma unroll
            for (int s = 0; s < 10; ++s) mk[s] = mv[4 * s + g4];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s = 0; s < 10; s += 2) {
                    t0 = fma(af[i][s], mk[s], t0);
                    t1 = fma(af[i][s + 1], mk[s + 1], t1);
                }
                const double t = wv_col_sum(t0 + t1);
                if (g4 == 0 && 16 * i + pi < D) basev[16 * i + pi] = t;
            }
        }
        wave_lds_fence();
        {
            const int l = lv;
            int f1, b1, b2;
            wv_ring(l, f1, b1, b2);
            const double base = basev[l] - b_l;
            const double f0 = (mv[f1] - mv[b2]) * mv[b1] - m_l + theta;   // drift at the mean
            const double Ef = Cxx + f0;
            wave_lds_fence();
            if (lane < D) {
                basev[l] = base;
                a.Ef[e * D + l] = Ef;
                if (a.dth_t) a.dth_t[e * D + l] = Ef + base;
                const double r0 = f0 + base;
                a0 = r0 * r0;
            }
        }
        wave_lds_fence();
    };
    bool ok = true;
    // diagonal block J of both steps by wavefront J & 1
#define WV_FACTOR_BOTH(J)                                                            \
    do {                                                                             \
        __syncthreads();                                                             \
        if (wv == ((J) & 1)) {                                                       \
            const int r = wv_factor_diag<J>(buf, zslot, other, lane);                \
            ok = ok && (r & 1);                                                      \
            if (lane == 0 && !(r & 2)) pd_flags[1 - wv] = 0;                         \
        } else if ((J) < 2) {                                                        \
            phase3();                                                                \
        }                                                                            \
        __syncthreads();                                                             \
    } while (0)
    WV_FACTOR_BOTH(0);
    WV_STAMP(27);
    {
        // block column 0: panel L_10, L_20 = S_I0 T_0'
        double t0[4], s10[4], s20[4];
        tjt4(0, t0);
        rows4(16 + pi, 0, s10);
        rows4(32 + pi, 0, s20);
        wv_f64x4 l10 = {0.0, 0.0, 0.0, 0.0}, l20 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            l10 = wv_mfma(s10[s], t0[s], l10);
            l20 = wv_mfma(s20[s], t0[s], l20);
        }
        wave_lds_fence();
        tile_wr(1, 0, l10);
        tile_wr(2, 0, l20);
        wave_lds_fence();
    }
    {
        // S_11 -= L_10 L_10',  S_21 -= L_20 L_10'
        double a10[4], a20[4], b10[4];
        rows4(16 + pi, 0, a10);
        rows4(32 + pi, 0, a20);
        rows4(16 + c16, 0, b10);
        wv_f64x4 s11, s21;
        tile_rd(1, 1, s11);
        tile_rd(2, 1, s21);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s11 = wv_mfma(-a10[s], b10[s], s11);
            s21 = wv_mfma(-a20[s], b10[s], s21);
        }
        wave_lds_fence();
        tile_wr(1, 1, s11);
        tile_wr(2, 1, s21);
        wave_lds_fence();
    }
    WV_STAMP(28);
    WV_RELANE();
    WV_FACTOR_BOTH(1);
    WV_STAMP(29);
    {
        // block column 1: panel L_21 = S_21 T_1'
        double t1[4], s21[4];
        tjt4(1, t1);
        rows4(32 + pi, 16, s21);
        wv_f64x4 l21 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) l21 = wv_mfma(s21[s], t1[s], l21);
        wave_lds_fence();
        tile_wr(2, 1, l21);
        wave_lds_fence();
    }
    {
        // S_22 -= L_20 L_20' + L_21 L_21'
        double a20[4], b20[4], a21[4], b21[4];
        rows4(32 + pi, 0, a20);
        rows4(32 + c16, 0, b20);
        rows4(32 + pi, 16, a21);
        rows4(32 + c16, 16, b21);
        wv_f64x4 s22, s22b = {0.0, 0.0, 0.0, 0.0};
        tile_rd(2, 2, s22);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s22 = wv_mfma(-a20[s], b20[s], s22);
            s22b = wv_mfma(-a21[s], b21[s], s22b);
        }
        s22 += s22b;
        wave_lds_fence();
        tile_wr(2, 2, s22);
        wave_lds_fence();
    }
    WV_STAMP(30);
    WV_RELANE();
    WV_FACTOR_BOTH(2);
#undef WV_FACTOR_BOTH
    ok = ok && __builtin_amdgcn_readfirstlane(pd_flags[wv]) != 0;

    WV_STAMP(18);
    WV_RELANE();
    // hp[j] = sum_d Sigma^-1_d (r+^2 + r-^2)/2 and hq[j] = sum_d Sigma^-1_d (r+^2 - r-^2)/4 of
    // sigma-point pair 16 j + c16, summed over this lane's state indices
    double hp[3] = {0.0, 0.0, 0.0}, hq[3] = {0.0, 0.0, 0.0};
    // (three vector values, not an array: the array form ended up in scratch memory, whose reloads
    // wait for the A fragments still in flight)
    wv_f64x4 mb0 = {0.0, 0.0, 0.0, 0.0}, mb1 = mb0, mb2 = mb0;
#define WV_MB(i) ((i) == 0 ? mb0 : ((i) == 1 ? mb1 : mb2))
    // ---- phase 3 has run inside the factorisation (see phase3 above); the dense <df/dx> if wanted ----
    if (a.Edf) {
        // Jacobian of the drift at the mean; row i is filled in the order
        // [i] = -1, [i+1] = m[i-1], [i-2] = -m[i-1], [i-1] = m[i+1] - m[i-2].
        double* E = a.Edf + e * DD;
#pragma unroll 5
        for (int q = lane; q < DD; q += 64) {
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
    if (!ok) {
        // Not positive definite: flag the step for the fallback pass that follows this
        // kernel and poison its outputs in case there is none.
        const double nan = __builtin_nan("");
        if (lane == 0) {
            a.Esde_t[e] = nan;
            if (a.status) a.status[e] = VGPA_ST_CHOL_FALLBACK | VGPA_ST_UNSUPPORTED;
        }
        if (lane < D) {
            a.dEdm[e * D + lane] = nan;
            if (a.dSig_t) a.dSig_t[e * D + lane] = nan;
        }
        for (int q = lane; q < SG; q += 64) a.dEdS[e * SG + q] = nan;
        return;
    }

    WV_STAMP(19);

    WV_RELANE();
    // ---- phase 4: G = A L, then the sigma-point residuals ---------------------------------
    // hp[j] = sum_d Sigma^-1_d (r+^2 + r-^2)/2 and hq[j] = sum_d Sigma^-1_d (r+^2 - r-^2)/4 of
    // sigma-point pair 16 j + c16, summed over this lane's state indices
    {
        // the B operands L'[4 s + g4][16 j + c16], s >= 4 j (L' is lower triangular: in the
        // diagonal blocks the entries above the diagonal are read as zero)
        double lb[3][10];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = 16 * j + c16;
#pragma unroll
            for (int s = 4 * j; s < 10; ++s) {
                const int k = 4 * s + g4;
                lb[j][s] = *((s < 4 * j + 4 && k < col) ? zslot : buf + k * LD + col);
            }
        }
        // one tile row of G at a time (the three tiles of a row share the A fragments, and
        // only 24 result registers are alive at once)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wv_f64x4 G[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                G[j] = wv_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 4 * j; s < 10; ++s) G[j] = wv_mfma(af[i][s], lb[j][s], G[j]);
            }
            const int d0r = 16 * i + 4 * g4;
            const bool rows_valid = d0r < D;
            const int d0 = rows_valid ? d0r : D - 4;
            // window rows d0 - 2 .. d0 + 4 on the ring: only the first lane group of tile row 0
            // (rows -2, -1) and the lanes with d0 = 36 (row 40) wrap
            int rw[7];
            double mm[7];
#pragma unroll
            for (int w = 0; w < 7; ++w) {
                rw[w] = d0 - 2 + w;
                mm[w] = mvx[d0 + w];
            }
            if (i == 0) {
                const int fix = (g4 == 0) ? D : 0;
                rw[0] += fix;
                rw[1] += fix;
            }
            if (i == 2) rw[6] -= (d0 == D - 4) ? D : 0;
            // per state index d = d0 + q (position q + 2 of the window): the drift at the mean
            // plus A m - b, and the two mean factors of the quadratic term
            double kq[4], aq[4], bq[4], si[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                aq[q] = mm[q + 3] - mm[q];
                bq[q] = mm[q + 1];
                kq[q] = fma(aq[q], bq[q], theta - mm[q + 2]) + basev[d0 + q];
                si[q] = rows_valid ? sIv[d0 + q] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wave_lds_fence();      // keeps the LDS reads of one tile together (register pressure)
                const int col = min(16 * j + c16, D - 1);
                const bool col_valid = 16 * j + c16 < D;
                double o[7];
#pragma unroll
                for (int w = 0; w < 7; ++w) {
                    o[w] = *((rw[w] >= col) ? buf + rw[w] * LD + col : zslot);      // L' is lower triangular
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // With x+- = m +- o the residuals of the two sigma points are
                    //   r+- = (a +- de)(b +- ep) - (m0 +- o0) + theta + base +- c G
                    // so  (r+ + r-)/2 = k + de ep   and   (r+ - r-)/2 = a ep + de b - o0 + G'  (G' = A L' = c G).
                    const double de = o[q + 3] - o[q], ep = o[q + 1];
                    const double sp = fma(de, ep, kq[q]);
                    const double dp = fma(aq[q], ep, fma(de, bq[q], G[j][q] - o[q + 2]));
                    const double t = fma(dp, dp, sp * sp);             // (r+^2 + r-^2) / 2
                    hp[j] = fma(si[q], t, hp[j]);
                    hq[j] = fma(si[q] * sp, dp, hq[j]);                // (r+^2 - r-^2) / 4
                    if (HYPER) {
                        const double add = col_valid ? 2.0 * w1 * t : 0.0;
                        if (i == 0) mb0[q] += add;
                        if (i == 1) mb1[q] += add;
                        if (i == 2) mb2[q] += add;
                    }
                }
            }
        }
    }
    WV_STAMP(21);
    WV_RELANE();
    // ---- phase 5: Esde(t), u, g ------------------------------------------------------------
    {
        double tot = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            hp[j] = wv_col_sum(hp[j]);
            hq[j] = wv_col_sum(hq[j]);
            tot += (16 * j + c16 < D) ? hp[j] : 0.0;
        }
        // sum over the sigma-point pairs, v+ + v- = 2 hp; the centre point: component l = lane
        tot = 2.0 * readlane_f64(wv_row_sum(tot), 0);
        const double v0 = readlane_f64(wv_row_sum(wv_col_sum(lane < D ? si_l * a0 : 0.0)), 0);
        const double Esde = 0.5 * (w0 * v0 + w1 * tot);
        if (g4 == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = 16 * j + c16;
                if (col < D) {
                    const double gk = c2 * (0.25 * hp[j] - 0.5 * Esde);   // c2 ((v+ + v-)/8 - Esde/2): W = sqrt(c2) W'
                    const double sq = sqrt(fabs(gk));
                    uv[col] = 4.0 * hq[j];                            // v+ - v-
                    sav[col] = sq;
                    sbv[col] = copysign(sq, gk);
                }
            }
        }
        if (lane == 0) {
            a.Esde_t[e] = Esde;
            if (a.status) a.status[e] = 0;
        }
        if (HYPER && a.dSig_t) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double rs = wv_row_sum(WV_MB(i)[q]);
                    if (i == 0) mb0[q] = rs;
                    if (i == 1) mb1[q] = rs;
                    if (i == 2) mb2[q] = rs;
                }
            // lane (g4, c16 == q) stores row 16 i + 4 g4 + q
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int d = 16 * i + 4 * g4 + c16;
                const wv_f64x4 mbi = WV_MB(i);
                double t = mbi[0];
                t = (c16 == 1) ? mbi[1] : t;
                t = (c16 == 2) ? mbi[2] : t;
                t = (c16 == 3) ? mbi[3] : t;
                // the centre point: w0 (f0 + base)^2 of component d, held by lane d
                const double ctr = __shfl(a0, min(d, D - 1), 64);
                if (c16 < 4 && 16 * i + 4 * g4 < D) a.dSig_t[e * D + d] = w0 * ctr + t;
            }
        }
    }
    wave_lds_fence();

    WV_STAMP(22);
    // ---- phase 6: W = L^-1 in place ---------------------------------------------------------
    //   W_10 = -T_1 (L_10 T_0)     W_21 = -T_2 (L_21 T_1)     W_20 = -T_2 (L_20 T_0 + L_21 W_10)
    {
        // B operand T_J: element [k][c] = T_J[k][c], zero above the diagonal
        auto tj4 = [&](int J, double (&o)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 4 * s + g4;
                o[s] = *((c16 <= k) ? buf + (16 * J + c16) * LD + 16 * J + k + 2 : zslot);
            }
        };
        // A operand -T_I: element [rho][k] = -T_I[pi][k] (NV valid rows / columns)
        auto neg_ti4 = [&](int I, int NV, double (&o)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 4 * s + g4;
                o[s] = -*((k <= pi && pi < NV) ? buf + (16 * I + k) * LD + 16 * I + pi + 2 : zslot);
            }
        };
        auto xs_wr = [&](const wv_f64x4& t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Xs[(4 * g4 + q) * LD + c16] = t[q];
        };
        auto xs_rd = [&](double (&o)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) o[s] = Xs[(4 * s + g4) * LD + c16];
        };
        const wv_f64x4 zero = {0.0, 0.0, 0.0, 0.0};
        double t0[4], t1[4], l10[4], l20[4], l21[4], nt1[4], nt2[4];
        tj4(0, t0);
        tj4(1, t1);
        rows4(16 + pi, 0, l10);
        rows4(32 + pi, 0, l20);
        rows4(32 + pi, 16, l21);
        neg_ti4(1, 16, nt1);
        neg_ti4(2, 8, nt2);
        wv_f64x4 x10 = zero, x20 = zero, x21 = zero;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x10 = wv_mfma(l10[s], t0[s], x10);
            x21 = wv_mfma(l21[s], t1[s], x21);
            x20 = wv_mfma(l20[s], t0[s], x20);
        }
        double xb[4], yb[4];
        wave_lds_fence();
        xs_wr(x10);
        wave_lds_fence();
        xs_rd(xb);
        wave_lds_fence();
        xs_wr(x21);
        wave_lds_fence();
        xs_rd(yb);
        wave_lds_fence();
        wv_f64x4 w10 = zero, w21 = zero;
#pragma unroll
        for (int s = 0; s < 4; ++s) w10 = wv_mfma(nt1[s], xb[s], w10);
#pragma unroll
        for (int s = 0; s < 2; ++s) w21 = wv_mfma(nt2[s], yb[s], w21);     // T_2 has 8 columns
        tile_wr(1, 0, w10);
        tile_wr(2, 1, w21);
        wave_lds_fence();
        double wb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wb[s] = buf[(16 + 4 * s + g4) * LD + c16];      // W_10, B operand
#pragma unroll
        for (int s = 0; s < 4; ++s) x20 = wv_mfma(l21[s], wb[s], x20);
        wave_lds_fence();
        xs_wr(x20);
        wave_lds_fence();
        xs_rd(xb);
        wv_f64x4 w20 = zero;
#pragma unroll
        for (int s = 0; s < 2; ++s) w20 = wv_mfma(nt2[s], xb[s], w20);
        wave_lds_fence();
        tile_wr(2, 0, w20);
        // diagonal blocks: W_JJ = T_J, from the shifted upper triangle to its place, with zeros above
        // the diagonal (the blocks of W above the diagonal blocks are never read)
        const int ra = lane >> 2, cb = 4 * (lane & 3);
        double t[3][4];
#pragma unroll
        for (int J = 0; J < 3; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = cb + q;
                const bool valid = (b <= ra) && (J < 2 || ra < 8);
                t[J][q] = valid ? buf[min(16 * J + b, D - 1) * LD + 16 * J + ra + 2] : 0.0;
            }
        wave_lds_fence();
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            double* dst = buf + (16 * J + ra) * LD + 16 * J + cb;
            *reinterpret_cast<double2*>(dst) = make_double2(t[J][0], t[J][1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(t[J][2], t[J][3]);
        }
        {
            double* dst = buf + ((ra < 8 && cb < 8) ? (32 + ra) * LD + 32 + cb : C::dump_quad(lane));
            *reinterpret_cast<double2*>(dst) = make_double2(t[2][0], t[2][1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(t[2][2], t[2][3]);
        }
    }
    wave_lds_fence();

    WV_STAMP(23);
    WV_STAMP(24);
    WV_RELANE();
    // ---- phase 7: dEsde/dm = (w c / 2) W' u and dEsde/dS = W' diag(g) W ----------------------
    // One pass over the rows k of W.  Columns 40..47 of a row are whatever follows in LDS
    // (finite numbers): they only reach the padded rows / columns of the results.
    {
        wv_f64x4 acc[3][3];
        double yv[3] = {0.0, 0.0, 0.0};       // W'u of column 16 t + pi, this lane group's rows k
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = wv_f64x4{0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const int k = 4 * s + g4;
            const double sa = sav[k], sb = sbv[k], uk = uv[k];
            const int nt = s / 4 + 1;                        // column blocks 0 .. nt-1 reach row k
            double ao[3], bo[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < nt) {
                    const double wa = buf[k * LD + 16 * t + pi];
                    yv[t] = fma(wa, uk, yv[t]);
                    ao[t] = sa * wa;
                    bo[t] = sb * buf[k * LD + 16 * t + c16];
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (i < nt && j <= i) acc[i][j] = wv_mfma(ao[i], bo[j], acc[i][j]);
            }
        }
        // dEsde/dm = (w c / 2) W' u, W = sqrt(c2) W': the first lane group holds row 16 t + pi
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double y = wv_col_sum(yv[t]);
            if (g4 == 0 && 16 * t + pi < D) a.dEdm[e * D + 16 * t + pi] = 0.5 * w1 * c2 * y;
        }
    WV_STAMP(25);
        double* Gd = a.dEdS + e * SG;
        if constexpr (PACKED_G) {
            // Row 16 i + x of the packed layout starts at base_i + width_i x.  Diagonal tiles go out
            // transposed like below (four consecutive doubles of row 16 i + c16); a tile below the diagonal
            // is stored once, as the rows it is: four 8-byte stores, each a full 128-byte line per
            // lane group.
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                constexpr int BASE[3] = {0, 256, 768};
                const int base = BASE[i], width = 16 * (i + 1);
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    if ((i < 2 || g4 < 2) && (j < 2 || c16 < 8)) {
                        if (j == i) {
                            double* dt_ = Gd + base + width * c16 + 16 * i + 4 * g4;
                            *reinterpret_cast<double2*>(dt_) = make_double2(acc[i][j][0], acc[i][j][1]);
                            *reinterpret_cast<double2*>(dt_ + 2) = make_double2(acc[i][j][2], acc[i][j][3]);
                        } else {
                            double* dst = Gd + base + width * 4 * g4 + 16 * j + c16;
#pragma unroll
                            for (int q = 0; q < 4; ++q) dst[q * width] = acc[i][j][q];
                        }
                    }
                }
            }
        } else {
            // The result is bitwise symmetric, so only the tiles on and below the diagonal are formed.
            // A lane holds rows 4 g4 .. 4 g4 + 3 of one column of a tile: stored transposed, that is
            // four consecutive doubles of row (16 j + c16) -- two 16-byte stores; tiles below the
            // diagonal are stored both ways.
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    if ((i < 2 || g4 < 2) && (j < 2 || c16 < 8)) {
                        double* dt_ = Gd + (16 * j + c16) * D + 16 * i + 4 * g4;
                        *reinterpret_cast<double2*>(dt_) = make_double2(acc[i][j][0], acc[i][j][1]);
                        *reinterpret_cast<double2*>(dt_ + 2) = make_double2(acc[i][j][2], acc[i][j][3]);
                        if (j < i) {
                            double* dst = Gd + (16 * i + 4 * g4) * D + 16 * j + c16;
#pragma unroll
                            for (int q = 0; q < 4; ++q) dst[q * D] = acc[i][j][q];
                        }
                    }
                }
        }
    }
    WV_STAMP(26);
#ifdef VGPA_STAMPS
    __builtin_amdgcn_s_waitcnt(0);            // how long the wavefront stays resident for its stores
    WV_STAMP(31);
#endif
#undef WV_RELANE
#undef WV_MB
}
