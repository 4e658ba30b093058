// Per-time-step expected SDE energy Esde(t) and its gradients w.r.t. m(t), S(t)
// for the five dynamical systems, gfx950.
//
// Every (problem, time step) pair is independent.  The 1-D and 3-D systems use
// one lane per pair; the Lorenz-96 unscented kernel uses one wavefront per pair
// with the covariance factor, its inverse and A(t) held in LDS.
//
// Unscented transform in closed form (D-dimensional Gaussian N(m, S), S = L L',
// c^2 = D + kappa, kappa = 1.05 D, sigma points m, m +/- c L[:,i], weights
// w0 = kappa / c^2, w = 1 / (2 c^2)):
//   r_k     = f(chi_k) + A chi_k - b,     v_k = sum_d SigInv_dd r_kd^2
//   Esde    = 1/2 sum_k w_k v_k
//   dEsde/dm = (w c / 2) L^-T (v+ - v-)
//   dEsde/dS = L^-T diag( (v+ + v-)/8 - Esde/2 ) L^-1
// which is what the per-sigma-point solves of core/gradEsde_mS.py:44-64 followed by
// dynamics/sys_lorenz_96_ut.py:238-241 evaluate to (the centre point has z = 0 and
// the Esde m terms cancel exactly).
#include "common.h"
#include "esde_args.h"
#include "sym_rows.h"
#include "linalg.h"

#include <cmath>
#include <type_traits>

namespace {

using vgpa::EsdeArgs;
using vgpa::readlane_f64;
using vgpa::wave_lds_fence;

// ---------------------------------------------------------------------------
// Ornstein-Uhlenbeck and double-well, D == 1: one lane per (problem, step).
// dynamics/sys_ornstein_uhlenbeck.py:133-153, dynamics/sys_double_well.py:139-181,
// Gaussian moments auxiliary/momGauss.py:47-86.
// ---------------------------------------------------------------------------
template <int SYSTEM>
__global__ void esde_1d_kernel(EsdeArgs a) {
    const size_t total = (size_t)a.P * a.N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e / a.N);
    const double theta = a.theta[(size_t)p * a.theta_stride];
    const double SigInv = a.sig_inv_diag[(size_t)p * a.sig_stride];
    const double A = a.A[e], b = a.b[e], m = a.m[e], S = a.S[e];
    const double m2 = m * m;
    const double Ex2 = m2 + S;
    if (SYSTEM == VGPA_SYS_OU) {
        const double tA = theta - A;
        const double Q1 = tA * tA;
        const double varQ = Ex2 * Q1 + 2 * m * tA * b + b * b;
        a.Esde_t[e] = 0.5 * SigInv * varQ;
        a.Ef[e] = -theta * m;
        a.Edf[e] = -theta;
        a.dEdm[e] = SigInv * (m * Q1 + theta * b - A * b);
        a.dEdS[e] = 0.5 * SigInv * Q1;
        if (a.dth_t) a.dth_t[e] = Ex2 * tA + m * b;
    } else {
        const double c = 4.0 * theta + A;
        const double c2 = c * c;
        const double m3 = m2 * m, m4 = m2 * m2, S2 = S * S;
        const double Ex3 = m3 + 3 * m * S;
        const double Ex4 = m4 + 6 * m2 * S + 3 * S2;
        const double Ex6 = m4 * m2 + 15 * m4 * S + 45 * m2 * S2 + 15 * S2 * S;
        const double varQ = 16.0 * Ex6 - 8.0 * c * Ex4 + 8.0 * b * Ex3 + c2 * Ex2 - 2.0 * b * c * m + b * b;
        a.Esde_t[e] = 0.5 * SigInv * varQ;
        a.Ef[e] = 4.0 * (theta * m - Ex3);
        a.Edf[e] = 4.0 * (theta - 3 * Ex2);
        const double Dm2 = 2 * m, DS2 = 1.0;
        const double Dm3 = 3 * Ex2, DS3 = 3 * m;
        const double Dm4 = 4 * Ex3, DS4 = 6 * Ex2;
        const double Dm6 = 6 * (m4 * m + 10 * m3 * S + 15 * m * S2);
        const double DS6 = 15 * m4 + 90 * m2 * S + 45 * S2;
        a.dEdm[e] = 0.5 * SigInv * (16.0 * Dm6 - 8.0 * c * Dm4 + 8.0 * b * Dm3 + c2 * Dm2 - 2.0 * b * c);
        a.dEdS[e] = 0.5 * SigInv * (16.0 * DS6 - 8.0 * c * DS4 + 8.0 * b * DS3 + c2 * DS2);
        if (a.dth_t) a.dth_t[e] = c * Ex2 - 4.0 * Ex4 - b * m;
    }
}

// ---------------------------------------------------------------------------
// Lorenz-63 helpers shared by the analytic and the unscented kernel.
// ---------------------------------------------------------------------------
struct L63In {
    double A[3][3], b[3], m[3], S[3][3];
    double vS, vR, vB;
    double iS[3];
};

// The 64 units of a wavefront are contiguous in every (P, N, RD) array.  When each lane reads its own RD
// numbers an instruction touches up to 36 cache lines; here the wavefront moves its 64 * RD doubles with
// 16-byte accesses of consecutive lanes and each lane picks its numbers out of LDS (odd stride RD: no
// bank conflicts), and the per-unit results go back the same way.  One buffer of 64 * 9 doubles per
// wavefront, used for one array at a time.  `wave_ok`: all 64 units of the wavefront exist.
struct L63Stage {
    static constexpr int BUF = 64 * 9;
    static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
    static __device__ __forceinline__ bool aligned(const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; }
    // v[0 .. RD) = src[(e0 + lane) * RD + ...], e0 = first unit of the wavefront
    template <int RD>
    static __device__ __forceinline__ void in(const double* __restrict__ src, size_t e0, int lane, double* buf, double (&v)[RD]) {
        constexpr int PIECES = 64 * RD / 2;
        const double2* s2 = reinterpret_cast<const double2*>(src + e0 * RD);
        double2* b2 = reinterpret_cast<double2*>(buf);
#pragma unroll
        for (int k = 0; k * 64 < PIECES; ++k)
            if (k * 64 + lane < PIECES) b2[k * 64 + lane] = s2[k * 64 + lane];
        fence();
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = buf[lane * RD + j];
        fence();
    }
    // the two halves of `in`: the wavefront's pieces into registers (no wait), and later through LDS
    template <int RD>
    static __device__ __forceinline__ void fetch(const double* __restrict__ src, size_t e0, int lane,
                                                 double (&pre)[2 * ((64 * RD / 2 + 63) / 64)]) {
        constexpr int PIECES = 64 * RD / 2, NI = (PIECES + 63) / 64;
        const double2* s2 = reinterpret_cast<const double2*>(src + e0 * RD);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const double2 t = s2[min(k * 64 + lane, PIECES - 1)];
            pre[2 * k] = t.x;
            pre[2 * k + 1] = t.y;
        }
    }
    template <int RD>
    static __device__ __forceinline__ void scatter(const double (&pre)[2 * ((64 * RD / 2 + 63) / 64)], int lane, double* buf,
                                                   double (&v)[RD]) {
        constexpr int PIECES = 64 * RD / 2, NI = (PIECES + 63) / 64;
        double2* b2 = reinterpret_cast<double2*>(buf);
#pragma unroll
        for (int k = 0; k < NI; ++k)
            if (k * 64 + lane < PIECES) b2[k * 64 + lane] = make_double2(pre[2 * k], pre[2 * k + 1]);
        fence();
#pragma unroll
        for (int j = 0; j < RD; ++j) v[j] = buf[lane * RD + j];
        fence();
    }
    template <int RD>
    static __device__ __forceinline__ void out(double* __restrict__ dst, size_t e0, int lane, double* buf, const double (&v)[RD]) {
        constexpr int PIECES = 64 * RD / 2;
#pragma unroll
        for (int j = 0; j < RD; ++j) buf[lane * RD + j] = v[j];
        fence();
        double2* d2 = reinterpret_cast<double2*>(dst + e0 * RD);
        const double2* b2 = reinterpret_cast<const double2*>(buf);
#pragma unroll
        for (int k = 0; k * 64 < PIECES; ++k)
            if (k * 64 + lane < PIECES) d2[k * 64 + lane] = b2[k * 64 + lane];
        fence();
    }
    // one per-unit array of RD doubles: staged when the wavefront is whole and the block is 16-byte aligned
    template <int RD>
    static __device__ __forceinline__ void load(const double* __restrict__ src, size_t e, int lane, bool wave_ok, double* buf,
                                                double (&v)[RD]) {
        if (wave_ok && aligned(src + (e - lane) * RD)) {
            in<RD>(src, e - lane, lane, buf, v);
        } else {
#pragma unroll
            for (int j = 0; j < RD; ++j) v[j] = src[e * RD + j];
        }
    }
    template <int RD>
    static __device__ __forceinline__ void store(double* __restrict__ dst, size_t e, int lane, bool wave_ok, double* buf,
                                                 const double (&v)[RD]) {
        if (wave_ok && aligned(dst + (e - lane) * RD)) {
            out<RD>(dst, e - lane, lane, buf, v);
        } else {
#pragma unroll
            for (int j = 0; j < RD; ++j) dst[e * RD + j] = v[j];
        }
    }
};

// per-wavefront staging context of the Lorenz-63 kernels
struct L63Io {
    double* buf;
    int lane;
    bool wave_ok;
};

__device__ inline void l63_load(const EsdeArgs& a, size_t e, int p, L63In& x, const L63Io& io) {
    double Af[9], Sf[9];
    const size_t e0 = e - io.lane;
    if (io.wave_ok && L63Stage::aligned(a.A + e0 * 9) && L63Stage::aligned(a.S + e0 * 9) &&
        L63Stage::aligned(a.b + e0 * 3) && L63Stage::aligned(a.m + e0 * 3)) {
        // all four arrays are requested before the first one is waited for
        double pa[10], ps[10], pb[4], pm[4];
        L63Stage::fetch<9>(a.A, e0, io.lane, pa);
        L63Stage::fetch<9>(a.S, e0, io.lane, ps);
        L63Stage::fetch<3>(a.b, e0, io.lane, pb);
        L63Stage::fetch<3>(a.m, e0, io.lane, pm);
        L63Stage::scatter<9>(pa, io.lane, io.buf, Af);
        L63Stage::scatter<9>(ps, io.lane, io.buf, Sf);
        L63Stage::scatter<3>(pb, io.lane, io.buf, x.b);
        L63Stage::scatter<3>(pm, io.lane, io.buf, x.m);
    } else {
        L63Stage::load<9>(a.A, e, io.lane, false, io.buf, Af);
        L63Stage::load<9>(a.S, e, io.lane, false, io.buf, Sf);
        L63Stage::load<3>(a.b, e, io.lane, false, io.buf, x.b);
        L63Stage::load<3>(a.m, e, io.lane, false, io.buf, x.m);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x.iS[i] = a.sig_inv_diag[(size_t)p * a.sig_stride + i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            x.A[i][j] = Af[i * 3 + j];
            x.S[i][j] = Sf[i * 3 + j];
        }
    }
    const double* th = a.theta + (size_t)p * a.theta_stride;
    x.vS = th[0];
    x.vR = th[1];
    x.vB = th[2];
}

// <f>, <df/dx> (dynamics/sys_lorenz_63.py:222-227) and <(f-g)' df/dtheta>
// (Efg_drift_theta, dynamics/sys_lorenz_63.py:491-549).
__device__ inline void l63_common_outputs(const EsdeArgs& a, size_t e, const L63In& x, const L63Io& io) {
    const double mx = x.m[0], my = x.m[1], mz = x.m[2];
    const double Ef[3] = {x.vS * (my - mx), x.vR * mx - my - x.S[2][0] - mx * mz, x.S[1][0] + mx * my - x.vB * mz};
    L63Stage::store<3>(a.Ef, e, io.lane, io.wave_ok, io.buf, Ef);
    if (a.Edf) {
        const double E[9] = {-x.vS, x.vS, 0.0, x.vR - mz, -1.0, -mx, my, mx, -x.vB};
        L63Stage::store<9>(a.Edf, e, io.lane, io.wave_ok, io.buf, E);
    }
    if (a.dth_t) {
        const double Sxx = x.S[0][0], Sxy = x.S[0][1], Sxz = x.S[0][2];
        const double Syy = x.S[1][1], Syz = x.S[1][2], Szz = x.S[2][2];
        const double Exx = Sxx + mx * mx, Exy = Sxy + mx * my, Eyy = Syy + my * my;
        const double Exz = Sxz + mx * mz, Ezz = Szz + mz * mz, Eyz = Syz + my * mz;
        const double Exxz = Sxx * mz + 2 * Sxz * mx + (mx * mx) * mz;
        const double Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz;
        const double A11 = x.A[0][0], A12 = x.A[0][1], A13 = x.A[0][2];
        const double A21 = x.A[1][0], A22 = x.A[1][1], A23 = x.A[1][2];
        const double A31 = x.A[2][0], A32 = x.A[2][1], A33 = x.A[2][2];
        const double dth[3] = {Eyy * (x.vS + A12) + Exx * (x.vS - A11) + Exy * (A11 - 2 * x.vS - A12) +
                                   A13 * (Eyz - Exz) + x.b[0] * (mx - my),
                               x.vR * Exx - Exy - Exxz + A21 * Exx + A22 * Exy + A23 * Exz - x.b[1] * mx,
                               -Exyz + x.vB * Ezz - A31 * Exz - A32 * Eyz - A33 * Ezz + x.b[2] * mz};
        L63Stage::store<3>(a.dth_t, e, io.lane, io.wave_ok, io.buf, dth);
    }
}

// ---------------------------------------------------------------------------
// Lorenz-63 with analytic Gaussian moments: one lane per (problem, step).
// Energy_dm_dS, dynamics/sys_lorenz_63.py:247-488.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) esde_l63_kernel(EsdeArgs a) {
    __shared__ __align__(16) double stage[4][L63Stage::BUF];
    const size_t total = (size_t)a.P * a.N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e / a.N);
    const int lane = threadIdx.x & 63;
    const L63Io io{stage[threadIdx.x >> 6], lane, (e - lane) + 64 <= total};
    L63In x;
    l63_load(a, e, p, x, io);
    l63_common_outputs(a, e, x, io);
    const double vS = x.vS, vR = x.vR, vB = x.vB;
    const double mx = x.m[0], my = x.m[1], mz = x.m[2];
    const double Sxx = x.S[0][0], Sxy = x.S[0][1], Sxz = x.S[0][2];
    const double Syy = x.S[1][1], Syz = x.S[1][2], Szz = x.S[2][2];
    const double A11 = x.A[0][0], A12 = x.A[0][1], A13 = x.A[0][2];
    const double A21 = x.A[1][0], A22 = x.A[1][1], A23 = x.A[1][2];
    const double A31 = x.A[2][0], A32 = x.A[2][1], A33 = x.A[2][2];
    const double b1 = x.b[0], b2 = x.b[1], b3 = x.b[2];
    // second to fourth order moments (:295-311)
    const double Exx = Sxx + mx * mx, Exy = Sxy + mx * my, Exz = Sxz + mx * mz;
    const double Eyy = Syy + my * my, Eyz = Syz + my * mz, Ezz = Szz + mz * mz;
    const double Exxy = Sxx * my + 2 * Sxy * mx + (mx * mx) * my;
    const double Exxz = Sxx * mz + 2 * Sxz * mx + (mx * mx) * mz;
    const double Exyy = Syy * mx + 2 * Sxy * my + (my * my) * mx;
    const double Exzz = Szz * mx + 2 * Sxz * mz + (mz * mz) * mx;
    const double Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz;
    const double Exxyy = Sxx * (my * my + Syy) + Syy * (mx * mx) + 4 * Sxy * mx * my + (mx * my) * (mx * my) + 2 * (Sxy * Sxy);
    const double Exxzz = Sxx * (mz * mz + Szz) + Szz * (mx * mx) + 4 * Sxz * mx * mz + (mx * mz) * (mx * mz) + 2 * (Sxz * Sxz);
    // <(f-g)^2> per component (:314-329)
    const double EX = (vS * vS) * (Eyy + Exx - 2 * Exy) + (A11 * A11) * Exx + (A12 * A12) * Eyy + (A13 * A13) * Ezz + b1 * b1 +
                      2 * (A11 * A12 * Exy + A11 * A13 * Exz - b1 * A11 * mx + A12 * A13 * Eyz - b1 * A12 * my - b1 * A13 * mz +
                           vS * (A11 * Exy + A12 * Eyy + A13 * Eyz - b1 * my - A11 * Exx - A12 * Exy - A13 * Exz + b1 * mx));
    const double EY = (vR * vR) * Exx + Eyy + Exxzz + (A21 * A21) * Exx + (A22 * A22) * Eyy + (A23 * A23) * Ezz + b2 * b2 +
                      2 * (Exyz - A21 * Exy - A22 * Eyy - A23 * Eyz - A21 * Exxz - A22 * Exyz - A23 * Exzz + A21 * A22 * Exy +
                           A21 * A23 * Exz + A22 * A23 * Eyz - vR * (Exy + Exxz - A21 * Exx - A22 * Exy - A23 * Exz) -
                           b2 * (vR * mx - my - Exz + A21 * mx + A22 * my + A23 * mz));
    const double EZ = Exxyy + (vB * vB) * Ezz + (A31 * A31) * Exx + (A32 * A32) * Eyy + (A33 * A33) * Ezz + b3 * b3 +
                      2 * (A31 * Exxy + A32 * Exyy + A33 * Exyz + A31 * A32 * Exy + A31 * A33 * Exz + A32 * A33 * Eyz -
                           vB * (Exyz + A31 * Exz + A32 * Eyz + A33 * Ezz) - b3 * (Exy - vB * mz + A31 * mx + A32 * my + A33 * mz));
    const double iSx = x.iS[0], iSy = x.iS[1], iSz = x.iS[2];
    a.Esde_t[e] = 0.5 * (iSx * EX + iSy * EY + iSz * EZ);
    if (a.dSig_t) {
        const double EXYZ[3] = {EX, EY, EZ};
        L63Stage::store<3>(a.dSig_t, e, io.lane, io.wave_ok, io.buf, EXYZ);
    }
    // moment derivatives w.r.t. the mean (:337-392); those equal to 1 are folded in
    const double dExx_dmx = 2 * mx, dExy_dmx = my, dExz_dmx = mz;
    const double dEyy_dmy = 2 * my, dExy_dmy = mx, dEyz_dmy = mz;
    const double dEzz_dmz = 2 * mz, dExz_dmz = mx, dEyz_dmz = my;
    const double dExxy_dmx = 2 * Exy, dExxz_dmx = 2 * Exz, dExyy_dmx = Eyy, dExzz_dmx = Ezz, dExyz_dmx = Eyz;
    const double dExxy_dmy = Exx, dExyy_dmy = 2 * Exy, dExyz_dmy = Exz;
    const double dExxz_dmz = Exx, dExzz_dmz = 2 * Exz, dExyz_dmz = Exy;
    const double dExxyy_dmx = 2 * Exyy, dExxzz_dmx = 2 * Exzz, dExxyy_dmy = 2 * Exxy, dExxzz_dmz = 2 * Exxz;
    // (:403-436)
    const double dmx1 = dExx_dmx * (vS * vS + A11 * A11) +
                        2 * (dExy_dmx * (-vS * vS + vS * A11 - vS * A12 + A11 * A12) + dExz_dmx * (A11 - vS) * A13 -
                             vS * A11 * dExx_dmx + b1 * (vS - A11));
    const double dmx2 = dExxzz_dmx + dExx_dmx * (vR * vR + A21 * A21) +
                        2 * (dExy_dmx * (-vR + vR * A22 - A21 + A21 * A22) + dExz_dmx * (vR * A23 + b2 + A21 * A23) +
                             dExyz_dmx * (1 - A22) - vR * dExxz_dmx + vR * A21 * dExx_dmx - A21 * dExxz_dmx -
                             A23 * dExzz_dmx - b2 * (vR + A21));
    const double dmx3 = dExxyy_dmx + (A31 * A31) * dExx_dmx +
                        2 * (dExy_dmx * (A31 * A32 - b3) + dExz_dmx * (A33 - vB) * A31 + dExyz_dmx * (A33 - vB) +
                             A31 * dExxy_dmx + A32 * dExyy_dmx - A31 * b3);
    const double dmy1 = dEyy_dmy * (vS * vS + A12 * A12) +
                        2 * (dExy_dmy * (-(vS * vS) + vS * A11 - vS * A12 + A11 * A12) + dEyz_dmy * (vS + A12) * A13 +
                             vS * A12 * dEyy_dmy - b1 * (vS + A12));
    const double dmy2 = dEyy_dmy * (1 + A22 * A22) +
                        2 * (dExy_dmy * (-vR + vR * A22 - A21 + A21 * A22) + dExyz_dmy * (1 - A22) - A22 * dEyy_dmy +
                             dEyz_dmy * (A22 * A23 - A23) + b2 * (1 - A22));
    const double dmy3 = dExxyy_dmy + (A32 * A32) * dEyy_dmy +
                        2 * (dExyz_dmy * (A33 - vB) + A31 * dExxy_dmy + A32 * dExyy_dmy + dExy_dmy * (A31 * A32 - b3) +
                             dEyz_dmy * (A33 - vB) * A32 - A32 * b3);
    const double dmz1 = (A13 * A13) * dEzz_dmz + 2 * (dEyz_dmz * (vS + A12) + dExz_dmz * (A11 - vS) - b1) * A13;
    const double dmz2 = dExxzz_dmz + (A23 * A23) * dEzz_dmz +
                        2 * (dExxz_dmz * (-vR - A21) + dExz_dmz * (vR * A23 + b2 + A21 * A23) + dExyz_dmz * (1 - A22) +
                             dEyz_dmz * (A22 * A23 - A23) - A23 * (dExzz_dmz + b2));
    const double dmz3 = dEzz_dmz * (vB * vB + A33 * A33) +
                        2 * ((A33 - vB) * (dExyz_dmz + dExz_dmz * A31 + dEyz_dmz * A32 - b3) - vB * A33 * dEzz_dmz);
    const double dEdm[3] = {0.5 * (dmx1 * iSx + dmx2 * iSy + dmx3 * iSz), 0.5 * (dmy1 * iSx + dmy2 * iSy + dmy3 * iSz),
                            0.5 * (dmz1 * iSx + dmz2 * iSy + dmz3 * iSz)};
    L63Stage::store<3>(a.dEdm, e, io.lane, io.wave_ok, io.buf, dEdm);
    // moment derivatives w.r.t. the covariance (:376-400) and dEsde/dS (:449-486)
    const double dExxy_dSxx = my, dExxz_dSxx = mz;
    const double dExxy_dSxy = 2 * mx, dExyy_dSxy = 2 * my, dExyz_dSxy = mz;
    const double dExzz_dSxz = 2 * mz, dExyz_dSxz = my, dExxz_dSxz = 2 * mx;
    const double dExyy_dSyy = mx, dExyz_dSyz = mx, dExzz_dSzz = mx;
    const double dExxyy_dSxx = Eyy, dExxzz_dSxx = Ezz, dExxyy_dSxy = 4 * Exy, dExxzz_dSxz = 4 * Exz;
    const double dExxyy_dSyy = Exx, dExxzz_dSzz = Exx;
    const double dSxx = iSx * ((vS - A11) * (vS - A11)) +
                        iSy * (dExxzz_dSxx + ((vR + A21) * (vR + A21)) - 2 * dExxz_dSxx * (vR + A21)) +
                        iSz * (dExxyy_dSxx + (A31 * A31) + 2 * A31 * dExxy_dSxx);
    const double dSxy = iSx * 2 * (vS * A11 - vS * vS - vS * A12 + A11 * A12) +
                        iSy * 2 * ((vR * A22 - vR - A21 + A21 * A22) + dExyz_dSxy * (1 - A22)) +
                        iSz * (dExxyy_dSxy + 2 * (dExyz_dSxy * (A33 - vB) + A31 * dExxy_dSxy + A32 * dExyy_dSxy + (A31 * A32 - b3)));
    const double dSxz = iSx * 2 * (A11 - vS) * A13 +
                        iSy * (dExxzz_dSxz + 2 * ((vR * A23 + b2 + A21 * A23) + dExyz_dSxz * (1 - A22) -
                                                  dExxz_dSxz * (vR + A21) - A23 * dExzz_dSxz)) +
                        iSz * 2 * ((A33 - vB) * A31 + dExyz_dSxz * (A33 - vB));
    const double dSyy = iSx * ((vS + A12) * (vS + A12)) + iSy * ((1 - A22) * (1 - A22)) +
                        iSz * (dExxyy_dSyy + (A32 * A32) + 2 * A32 * dExyy_dSyy);
    const double dSyz = iSx * 2 * (vS + A12) * A13 + iSy * 2 * (dExyz_dSyz * (1 - A22) + (A22 - 1) * A23) +
                        iSz * 2 * (dExyz_dSyz * (A33 - vB) + (A33 - vB) * A32);
    const double dSzz = iSx * (A13 * A13) + iSy * (dExxzz_dSzz + (A23 * A23) - 2 * A23 * dExzz_dSzz) +
                        iSz * ((vB - A33) * (vB - A33));
    const double G[9] = {0.5 * dSxx, 0.5 * dSxy, 0.5 * dSxz, 0.5 * dSxy, 0.5 * dSyy, 0.5 * dSyz,
                         0.5 * dSxz, 0.5 * dSyz, 0.5 * dSzz};
    L63Stage::store<9>(a.dEdS, e, io.lane, io.wave_ok, io.buf, G);
}

// ---------------------------------------------------------------------------
// Lorenz-63 with the unscented transform: one lane per (problem, step), all
// seven sigma points in registers.  This kernel follows the per-sample form of
// core/gradEsde_mS.py:44-64 with an explicit 3x3 inverse, so it also covers the
// diagonal Cholesky fallback of auxiliary/numerical.py:199-202.
// dynamics/sys_lorenz_63_ut.py:218-255.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) esde_l63_ut_kernel(EsdeArgs a) {
    __shared__ __align__(16) double stage[4][L63Stage::BUF];
    const size_t total = (size_t)a.P * a.N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e / a.N);
    const int lane = threadIdx.x & 63;
    const L63Io io{stage[threadIdx.x >> 6], lane, (e - lane) + 64 <= total};
    L63In x;
    l63_load(a, e, p, x, io);
    l63_common_outputs(a, e, x, io);
    constexpr int D = 3;
    const double kappa = 1.05 * D;
    const double c2 = D + kappa;
    int st = 0;
    // U = chol(c2 * S)' : offsets of sigma point i are column i of the lower factor.
    double L[3][3] = {};
    bool ok = true;
    {
        const double s00 = c2 * x.S[0][0], s10 = c2 * x.S[1][0], s11 = c2 * x.S[1][1];
        const double s20 = c2 * x.S[2][0], s21 = c2 * x.S[2][1], s22 = c2 * x.S[2][2];
        if (!(s00 > 0.0)) ok = false;
        if (ok) {
            L[0][0] = sqrt(s00);
            L[1][0] = s10 / L[0][0];
            L[2][0] = s20 / L[0][0];
            const double d1 = s11 - L[1][0] * L[1][0];
            if (!(d1 > 0.0)) ok = false;
            if (ok) {
                L[1][1] = sqrt(d1);
                L[2][1] = (s21 - L[2][0] * L[1][0]) / L[1][1];
                const double d2 = s22 - L[2][0] * L[2][0] - L[2][1] * L[2][1];
                if (!(d2 > 0.0)) ok = false;
                if (ok) L[2][2] = sqrt(d2);
            }
        }
    }
    if (!ok) {
        st |= VGPA_ST_CHOL_FALLBACK;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) L[i][j] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!(x.S[i][i] > 0.0)) st |= VGPA_ST_CHOL_FAILED;
            L[i][i] = sqrt(x.S[i][i]);
        }
    }
    // explicit inverse of S (adjugate)
    double Si[3][3];
    {
        const double (*S)[3] = x.S;
        const double c00 = S[1][1] * S[2][2] - S[1][2] * S[2][1];
        const double c01 = S[1][2] * S[2][0] - S[1][0] * S[2][2];
        const double c02 = S[1][0] * S[2][1] - S[1][1] * S[2][0];
        const double det = S[0][0] * c00 + S[0][1] * c01 + S[0][2] * c02;
        const double id = 1.0 / det;
        Si[0][0] = c00 * id;
        Si[1][0] = c01 * id;
        Si[2][0] = c02 * id;
        Si[0][1] = (S[0][2] * S[2][1] - S[0][1] * S[2][2]) * id;
        Si[1][1] = (S[0][0] * S[2][2] - S[0][2] * S[2][0]) * id;
        Si[2][1] = (S[0][1] * S[2][0] - S[0][0] * S[2][1]) * id;
        Si[0][2] = (S[0][1] * S[1][2] - S[0][2] * S[1][1]) * id;
        Si[1][2] = (S[0][2] * S[1][0] - S[0][0] * S[1][2]) * id;
        Si[2][2] = (S[0][0] * S[1][1] - S[0][1] * S[1][0]) * id;
    }
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);
    double mbar[3] = {0, 0, 0}, dm[3] = {0, 0, 0}, dS[3][3] = {};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double z[3] = {0, 0, 0};
        if (k >= 1) {
            const int i = (k - 1) % 3;
            const double sgn = (k <= 3) ? 1.0 : -1.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) z[r] = sgn * L[r][i];
        }
        const double wk = (k == 0) ? w0 : w1;
        double xk[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) xk[r] = x.m[r] + z[r];
        double f[3];
        f[0] = -x.vS * (xk[0] - xk[1]);
        f[1] = (x.vR - xk[2]) * xk[0] - xk[1];
        f[2] = xk[0] * xk[1] - x.vB * xk[2];
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double res = f[r] + x.A[r][0] * xk[0] + x.A[r][1] * xk[1] + x.A[r][2] * xk[2] - x.b[r];
            const double r2 = res * res;
            mbar[r] += wk * r2;
            v += x.iS[r] * r2;
        }
        double u[3], uz[3], zs[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            u[r] = Si[r][0] * xk[0] + Si[r][1] * xk[1] + Si[r][2] * xk[2];    // S^-1 x_k
            uz[r] = Si[r][0] * z[0] + Si[r][1] * z[1] + Si[r][2] * z[2];      // S^-1 z_k
            zs[r] = z[0] * Si[0][r] + z[1] * Si[1][r] + z[2] * Si[2][r];      // z_k' S^-1
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dm[r] += wk * (0.5 * v * u[r]);
#pragma unroll
            for (int c = 0; c < 3; ++c) dS[r][c] += wk * (0.5 * v * uz[r] * zs[c]);
        }
    }
    const double Esde = 0.5 * (x.iS[0] * mbar[0] + x.iS[1] * mbar[1] + x.iS[2] * mbar[2]);
    a.Esde_t[e] = Esde;
    if (a.dSig_t) L63Stage::store<3>(a.dSig_t, e, io.lane, io.wave_ok, io.buf, mbar);
    double dEdm[3], dEdS[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double sim = Si[r][0] * x.m[0] + Si[r][1] * x.m[1] + Si[r][2] * x.m[2];
        dEdm[r] = dm[r] - Esde * sim;
#pragma unroll
        for (int c = 0; c < 3; ++c) dEdS[r * 3 + c] = 0.5 * (dS[r][c] - Esde * Si[r][c]);
    }
    L63Stage::store<3>(a.dEdm, e, io.lane, io.wave_ok, io.buf, dEdm);
    L63Stage::store<9>(a.dEdS, e, io.lane, io.wave_ok, io.buf, dEdS);
    if (a.status) a.status[e] = st;
}

// ---------------------------------------------------------------------------
// Lorenz-96 with the unscented transform: one wavefront per (problem, step).
// dynamics/sys_lorenz_96_ut.py:217-247; drift :17-40, <f> :260-291, <df/dx> :293-330.
// LDS holds (rows padded by one double): Lm = S then its Cholesky factor then
// diag(g) L^-1; Am = A(t) then L^-1; Rp / Rm = per-component weighted squared
// residuals of the +/- sigma points.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// Lorenz-96, one step whose S(t) is not positive definite.  The reference then
// takes sigma points from the square roots of the diagonal alone, unscaled
// (auxiliary/numerical.py:201-202), and its gradient code solves with the full
// S(t) (core/gradEsde_mS.py:48-60).  With z_i = +/- s_i e_i, s_i = sqrt(S_ii), and
// Si = inv(S) that gives
//   dEsde/dm = (w/2) Si (s .* (v+ - v-))
//   dEsde/dS = (w/4) Si diag(S_ii (v+ + v-)) Si - (Esde_t/2) Si .
// One wavefront.  <f>, <df/dx> and the theta integrand do not depend on the factorisation; they
// are written here as well, so that a caller may hand over a step it has not touched.
// LDS: 3 D (D+1) + 5 D doubles.
// ---------------------------------------------------------------------------
__device__ void l96_fallback_step(const EsdeArgs& a, size_t e, double* lds) {
    const int D = a.D, LD = D + 1, MS = D * LD;
    const int l = threadIdx.x;
    const int p = (int)(e / a.N);
    const size_t DD = (size_t)D * D;
    double* Sm = lds;
    double* Si = lds + MS;
    double* Am = lds + 2 * MS;
    double* mv = lds + 3 * MS;
    double* bv = mv + D;
    double* sv = bv + D;
    double* uv = sv + D;
    double* gv = uv + D;
    __syncthreads();
    const double theta = a.theta[(size_t)p * a.theta_stride];
    // packed (D = 40 only): S (bit 0) and dEsde/dS (bit 1) in the layout of sym_rows.h
    const bool packed_s = (a.packed & 1) != 0, packed_g = (a.packed & 2) != 0;
    const size_t SS = packed_s ? (size_t)vgpa::SymRows40::SIZE : DD;
    const size_t SG = packed_g ? (size_t)vgpa::SymRows40::SIZE : DD;
    const double* Ag = a.A + e * DD;
    const double* Sg = a.S + e * SS;
    for (int q = l; q < (int)DD; q += 64) {
        const int o = (q / D) * LD + (q % D);
        Am[o] = Ag[q];
        if (!packed_s) Sm[o] = Sg[q];
    }
    if (packed_s) {
        for (int q = l; q < (int)SS; q += 64) {
            int r, c;
            vgpa::SymRows40::coords(q, r, c);
            if (c < D) {
                const double v = Sg[q];
                Sm[r * LD + c] = v;
                if (c < 16 * (r / 16)) Sm[c * LD + r] = v;       // left of the diagonal block: the mirror image too
            }
        }
    }
    const bool act = l < D;
    double sI = 0.0, m_l = 0.0, sdiag = 1.0;
    if (act) {
        m_l = a.m[e * D + l];
        mv[l] = m_l;
        bv[l] = a.b[e * D + l];
        sI = a.sig_inv_diag[(size_t)p * a.sig_stride + l];
        sdiag = packed_s ? Sg[vgpa::SymRows40::row_offset(l) + l] : Sg[(size_t)l * D + l];
    }
    const double nan = __builtin_nan("");
    if (!__all(sdiag > 0.0)) {
        // the diagonal has no Cholesky factor either: numpy raises LinAlgError here
        if (l == 0) {
            a.Esde_t[e] = nan;
            if (a.status) a.status[e] = VGPA_ST_CHOL_FALLBACK | VGPA_ST_CHOL_FAILED;
        }
        if (act) {
            a.dEdm[e * D + l] = nan;
            if (a.dSig_t) a.dSig_t[e * D + l] = nan;
        }
        for (int q = l; q < (int)SG; q += 64) a.dEdS[e * SG + q] = nan;
        return;
    }
    if (act) sv[l] = sqrt(sdiag);
    __syncthreads();

    const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + 2 * D - 2) % D;
    double base = 0.0;
    if (act) {
        double am = 0.0;
        for (int k = 0; k < D; ++k) am = fma(Am[l * LD + k], mv[k], am);
        base = am - bv[l];
        const double Cxx = Sm[f1 * LD + b1] - Sm[b2 * LD + b1];
        const double Ef = Cxx + (mv[f1] - mv[b2]) * mv[b1] - m_l + theta;
        a.Ef[e * D + l] = Ef;
        if (a.dth_t) a.dth_t[e * D + l] = Ef + base;
    }
    if (a.Edf) {
        double* E = a.Edf + e * DD;
        for (int q = l; q < (int)DD; q += 64) {
            const int i = q / D, j = q % D;
            const int jf1 = (i + 1) % D, jb1 = (i + D - 1) % D, jb2 = (i + 2 * D - 2) % D;
            double v = 0.0;
            if (j == i) v = -1.0;
            if (j == jf1) v = mv[jb1];
            if (j == jb2) v = -mv[jb1];
            if (j == jb1) v = mv[jf1] - mv[jb2];
            E[q] = v;
        }
    }
    const double kappa = 1.05 * D;
    const double c2 = D + kappa;
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);
    double mbar = 0.0, a0 = 0.0;
    if (act) {
        const double f0 = (mv[f1] - mv[b2]) * mv[b1] - m_l + theta;
        const double r0 = f0 + base;
        mbar = w0 * r0 * r0;
        a0 = sI * r0 * r0;
    }
    const double v0 = wave_sum(a0);
    double vsum = 0.0, u_l = 0.0, g_l = 0.0;
    for (int i = 0; i < D; ++i) {
        const double si = sv[i];
        double ap = 0.0, am = 0.0;
        if (act) {
            const double sa = si * Am[l * LD + i];
            const double o0 = (l == i) ? si : 0.0, of1 = (f1 == i) ? si : 0.0;
            const double ob1 = (b1 == i) ? si : 0.0, ob2 = (b2 == i) ? si : 0.0;
            const double xf1 = mv[f1], xb1 = mv[b1], xb2 = mv[b2];
            const double fp = ((xf1 + of1) - (xb2 + ob2)) * (xb1 + ob1) - (m_l + o0) + theta;
            const double fm = ((xf1 - of1) - (xb2 - ob2)) * (xb1 - ob1) - (m_l - o0) + theta;
            const double rp = fp + base + sa;
            const double rm = fm + base - sa;
            mbar += w1 * (rp * rp + rm * rm);
            ap = sI * rp * rp;
            am = sI * rm * rm;
        }
        const double vp = wave_sum(ap), vm = wave_sum(am);
        vsum += vp + vm;
        if (l == i) {
            u_l = si * (vp - vm);
            g_l = si * si * (vp + vm);
        }
    }
    const double Esde = 0.5 * (w0 * v0 + w1 * vsum);
    if (l == 0) {
        a.Esde_t[e] = Esde;
        if (a.status) a.status[e] = VGPA_ST_CHOL_FALLBACK;
    }
    if (act) {
        if (a.dSig_t) a.dSig_t[e * D + l] = mbar;
        uv[l] = u_l;
        gv[l] = g_l;
    }
    __syncthreads();
    vgpa::gj_inverse(Sm, Si, D, LD, l);
    if (act) {
        double y = 0.0;
        for (int j = 0; j < D; ++j) y = fma(Si[l * LD + j], uv[j], y);
        a.dEdm[e * D + l] = 0.5 * w1 * y;
    }
    // The pivoted inverse is symmetric to rounding only; the lower triangle is computed and
    // mirrored, so that dEsde/dS is exactly symmetric like the output of the regular steps
    // (the backward scan's one-product form relies on that).
    double* G = a.dEdS + e * SG;
    for (int i = 0; i < D; ++i) {
        if (act && l <= i) {
            double s = 0.0;
            for (int k = 0; k < D; ++k) s = fma(Si[i * LD + k] * gv[k], Si[k * LD + l], s);
            const double v = 0.25 * w1 * s - 0.5 * Esde * Si[i * LD + l];
            if (packed_g) {
                G[vgpa::SymRows40::row_offset(i) + l] = v;
                if (vgpa::SymRows40::kept(l, i)) G[vgpa::SymRows40::row_offset(l) + i] = v;   // same diagonal block
            } else {
                G[i * D + l] = v;
                G[(size_t)l * D + i] = v;
            }
        }
    }
}

// After a kernel that only flags such steps: every wavefront looks at 64
// consecutive steps and redoes the flagged ones.
__global__ void __launch_bounds__(64) esde_l96_fixup_kernel(EsdeArgs a, size_t total) {
    extern __shared__ double lds[];
    const size_t e0 = (size_t)blockIdx.x * 64;
    const size_t mine = e0 + threadIdx.x;
    // flagged steps: by the status word, or, when the caller gave no status array, by the NaN
    // the energy kernel left in Esde_t (a step that is NaN for another reason stays NaN)
    bool flagged = false;
    if (mine < total) flagged = a.status ? (a.status[mine] & VGPA_ST_UNSUPPORTED) != 0 : isnan(a.Esde_t[mine]);
    unsigned long long todo = __ballot(flagged);
    while (todo) {
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        l96_fallback_step(a, e0 + i, lds);
    }
}

__global__ void __launch_bounds__(64) esde_l96_ut_kernel(EsdeArgs a) {
    extern __shared__ double lds[];
    const int D = a.D, LD = D + 1, MS = D * LD;
    const int l = threadIdx.x;
    const size_t e = blockIdx.x;            // p * N + t
    const int p = (int)(e / a.N);
    const size_t DD = (size_t)D * D;
    double* Lm = lds;
    double* Am = lds + MS;
    double* Rp = lds + 2 * MS;
    double* Rm = lds + 3 * MS;
    double* mv = lds + 4 * MS;
    double* bv = mv + D;
    double* vp = bv + D;
    double* vm = vp + D;
    double* gd = vm + D;

    const double theta = a.theta[(size_t)p * a.theta_stride];
    const double* Ag = a.A + e * DD;
    const double* Sg = a.S + e * DD;
    for (int q = l; q < (int)DD; q += 64) {
        const int o = (q / D) * LD + (q % D);
        Am[o] = Ag[q];
        Lm[o] = Sg[q];
    }
    const bool act = l < D;
    double sI = 0.0, m_l = 0.0;
    if (act) {
        m_l = a.m[e * D + l];
        mv[l] = m_l;
        bv[l] = a.b[e * D + l];
        sI = a.sig_inv_diag[(size_t)p * a.sig_stride + l];
    }
    __syncthreads();

    const int f1 = (l + 1) % D, b1 = (l + D - 1) % D, b2 = (l + 2 * D - 2) % D;
    double base = 0.0;   // (A m - b)[l]
    if (act) {
        double am = 0.0;
        for (int k = 0; k < D; ++k) am = fma(Am[l * LD + k], mv[k], am);
        base = am - bv[l];
        const double Cxx = Lm[f1 * LD + b1] - Lm[b2 * LD + b1];
        const double Ef = Cxx + (mv[f1] - mv[b2]) * mv[b1] - m_l + theta;
        a.Ef[e * D + l] = Ef;
        if (a.dth_t) a.dth_t[e * D + l] = Ef + base;
    }
    if (a.Edf) {
        // Jacobian of the drift at the mean; row i is filled in the order
        // [i] = -1, [i+1] = m[i-1], [i-2] = -m[i-1], [i-1] = m[i+1] - m[i-2].
        double* E = a.Edf + e * DD;
        for (int q = l; q < (int)DD; q += 64) {
            const int i = q / D, j = q % D;
            const int jf1 = (i + 1) % D, jb1 = (i + D - 1) % D, jb2 = (i + 2 * D - 2) % D;
            double v = 0.0;
            if (j == i) v = -1.0;
            if (j == jf1) v = mv[jb1];
            if (j == jb2) v = -mv[jb1];
            if (j == jb1) v = mv[jf1] - mv[jb2];
            E[q] = v;
        }
    }
    __syncthreads();

    // In-place Cholesky of S (lower triangle), lane = row.
    bool ok = true;
    for (int j = 0; j < D; ++j) {
        const double djj = Lm[j * LD + j];
        if (!(djj > 0.0)) { ok = false; break; }
        const double d = sqrt(djj);
        const double rd = 1.0 / d;
        double lij = 0.0;
        if (act && l >= j) {
            lij = (l == j) ? d : Lm[l * LD + j] * rd;
            Lm[l * LD + j] = lij;
        }
        __syncthreads();
        for (int k = j + 1; k < D; ++k) {
            const double lkj = Lm[k * LD + j];
            if (act && l >= k) Lm[l * LD + k] = fma(-lij, lkj, Lm[l * LD + k]);
        }
        __syncthreads();
    }
    if (!ok) {
        l96_fallback_step(a, e, lds);
        return;
    }
    // Zero the strict upper triangle so that column reads need no masking.
    if (act) {
        for (int k = l + 1; k < D; ++k) Lm[l * LD + k] = 0.0;
    }
    __syncthreads();

    const double kappa = 1.05 * D;
    const double c2 = D + kappa;
    const double c = sqrt(c2);
    const double w0 = kappa / c2, w1 = 1.0 / (2.0 * c2);

    // Centre point.
    double mbar = 0.0;   // sum_k w_k r_k[l]^2
    double a0 = 0.0;
    if (act) {
        const double f0 = (mv[f1] - mv[b2]) * mv[b1] - m_l + theta;
        const double r0 = f0 + base;
        mbar = w0 * r0 * r0;
        a0 = sI * r0 * r0;
    }
    const double v0 = wave_sum(a0);

    // Sigma points +/- column i of c L.
    for (int i = 0; i < D; ++i) {
        if (act) {
            double G = 0.0;   // (A L)[l][i]
            for (int j = i; j < D; ++j) G = fma(Am[l * LD + j], Lm[j * LD + i], G);
            const double cG = c * G;
            const double o0 = c * Lm[l * LD + i], of1 = c * Lm[f1 * LD + i];
            const double ob1 = c * Lm[b1 * LD + i], ob2 = c * Lm[b2 * LD + i];
            const double xf1 = mv[f1], xb1 = mv[b1], xb2 = mv[b2];
            const double fp = ((xf1 + of1) - (xb2 + ob2)) * (xb1 + ob1) - (m_l + o0) + theta;
            const double fm = ((xf1 - of1) - (xb2 - ob2)) * (xb1 - ob1) - (m_l - o0) + theta;
            const double rp = fp + base + cG;
            const double rm = fm + base - cG;
            const double rp2 = rp * rp, rm2 = rm * rm;
            mbar += w1 * (rp2 + rm2);
            Rp[i * LD + l] = sI * rp2;
            Rm[i * LD + l] = sI * rm2;
        }
    }
    __syncthreads();
    double vpl = 0.0, vml = 0.0;
    if (act) {
        for (int d = 0; d < D; ++d) {
            vpl += Rp[l * LD + d];
            vml += Rm[l * LD + d];
        }
    }
    const double Esde = 0.5 * (w0 * v0 + w1 * wave_sum(vpl + vml));
    if (l == 0) {
        a.Esde_t[e] = Esde;
        if (a.status) a.status[e] = 0;
    }
    if (act && a.dSig_t) a.dSig_t[e * D + l] = mbar;
    __syncthreads();

    // W = L^-1 (lower) into Am, lane = column, rows in sequence.
    double* Wm = Am;
    if (act) {
        for (int k = 0; k < D; ++k) Wm[k * LD + l] = 0.0;
    }
    __syncthreads();
    for (int i = 0; i < D; ++i) {
        const double rdi = 1.0 / Lm[i * LD + i];
        if (act && l <= i) {
            double s = 0.0;
            for (int k = 0; k < i; ++k) s = fma(Lm[i * LD + k], Wm[k * LD + l], s);
            Wm[i * LD + l] = (l == i) ? rdi : -s * rdi;
        }
        __syncthreads();
    }
    if (act) {
        vp[l] = vpl - vml;                            // u = v+ - v-
        gd[l] = 0.125 * (vpl + vml) - 0.5 * Esde;     // diagonal weights of dEsde/dS
    }
    __syncthreads();
    // dEsde/dm = (w1 c / 2) W' u
    if (act) {
        double y = 0.0;
        for (int k = l; k < D; ++k) y = fma(Wm[k * LD + l], vp[k], y);
        a.dEdm[e * D + l] = 0.5 * w1 * c * y;
        // V = diag(g) W into Lm (the factor itself is no longer needed)
        for (int k = 0; k < D; ++k) Lm[k * LD + l] = gd[k] * Wm[k * LD + l];
    }
    __syncthreads();
    // dEsde/dS[i][j] = sum_{k >= max(i,j)} V[k][i] W[k][j]; lane = j, rows i in sequence.
    double* G = a.dEdS + e * DD;
    for (int i = 0; i < D; ++i) {
        if (act) {
            double s = 0.0;
            for (int k = i; k < D; ++k) s = fma(Lm[k * LD + i], Wm[k * LD + l], s);
            G[i * D + l] = s;
        }
    }
}

#include "esde_l96_tiled.h"

template <int D, bool HYPER>
void launch_l96_tiled_h(const EsdeArgs& a, size_t total, hipStream_t st) {
    const size_t lds = sizeof(double) * L96Tiled<D, HYPER>::LDS_DOUBLES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(esde_l96_ut_tiled_kernel<D, HYPER>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((esde_l96_ut_tiled_kernel<D, HYPER>), dim3((unsigned)total), dim3(256), lds, st, a);
}

template <int D>
void launch_l96_tiled(const EsdeArgs& a, size_t total, hipStream_t st) {
    if constexpr (D == 40) vgpa::launch_esde_l96_wave40(a, st);
    else if (a.dSig_t) launch_l96_tiled_h<D, true>(a, total, st);
    else launch_l96_tiled_h<D, false>(a, total, st);
    {
        // steps the energy kernel flagged as not positive definite
        const size_t lds = sizeof(double) * ((size_t)3 * D * (D + 1) + 5 * D);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(esde_l96_fixup_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(esde_l96_fixup_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), lds, st, a, total);
    }
}

int launch_esde(int system, const EsdeArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.P * a.N;
    if (total == 0) return VGPA_OK;
    const int bs = 256;
    const unsigned grid = (unsigned)((total + bs - 1) / bs);
    switch (system) {
        case VGPA_SYS_OU:
            hipLaunchKernelGGL(esde_1d_kernel<VGPA_SYS_OU>, dim3(grid), dim3(bs), 0, st, a);
            break;
        case VGPA_SYS_DW:
            hipLaunchKernelGGL(esde_1d_kernel<VGPA_SYS_DW>, dim3(grid), dim3(bs), 0, st, a);
            break;
        case VGPA_SYS_L63:
            hipLaunchKernelGGL(esde_l63_kernel, dim3(grid), dim3(bs), 0, st, a);
            break;
        case VGPA_SYS_L63_UT:
            hipLaunchKernelGGL(esde_l63_ut_kernel, dim3(grid), dim3(bs), 0, st, a);
            break;
        case VGPA_SYS_L96_UT: {
            if (a.D == 40) { launch_l96_tiled<40>(a, total, st); break; }
            if (a.D == 16) { launch_l96_tiled<16>(a, total, st); break; }
            const size_t lds = sizeof(double) * ((size_t)4 * a.D * (a.D + 1) + 5 * a.D);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(esde_l96_ut_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(esde_l96_ut_kernel, dim3((unsigned)total), dim3(64), lds, st, a);
            break;
        }
        default:
            return vgpa::fail(VGPA_EINVAL, "vgpa_esde: unknown system %d", system);
    }
    return vgpa::check_launch("vgpa_esde");
}

}  // namespace

extern "C" int vgpa_esde(int system, int P, int N, int D, const double* A, const double* b, const double* m,
                         const double* S, const double* sig_inv_diag, int sig_stride, const double* theta,
                         int theta_stride, double* Esde_t, double* Ef, double* Edf, double* dEsde_dm,
                         double* dEsde_dS, double* dth_t, double* dSig_t, int* status, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1, "vgpa_esde: bad sizes P=%d N=%d D=%d", P, N, D);
    VGPA_REQUIRE(A && b && m && S && sig_inv_diag && theta && Esde_t && Ef && dEsde_dm && dEsde_dS,
                 "vgpa_esde: null pointer argument");
    VGPA_REQUIRE((size_t)P * N < 0x7fffffffull, "vgpa_esde: P*N too large");
    switch (system) {
        case VGPA_SYS_OU:
        case VGPA_SYS_DW:
            VGPA_REQUIRE(D == 1, "vgpa_esde: system %d needs D == 1 (got %d)", system, D);
            VGPA_REQUIRE(Edf, "vgpa_esde: Edf is required for D == 1");
            break;
        case VGPA_SYS_L63:
        case VGPA_SYS_L63_UT:
            VGPA_REQUIRE(D == 3, "vgpa_esde: Lorenz-63 needs D == 3 (got %d)", D);
            break;
        case VGPA_SYS_L96_UT:
            VGPA_REQUIRE(D >= 2, "vgpa_esde: Lorenz-96 needs D >= 2 (got %d)", D);
            if (D > VGPA_MAX_D) return vgpa::fail(VGPA_EUNSUPPORTED, "vgpa_esde: D=%d exceeds VGPA_MAX_D=%d", D, VGPA_MAX_D);
            break;
        default:
            return vgpa::fail(VGPA_EINVAL, "vgpa_esde: unknown system %d", system);
    }
    EsdeArgs a{P, N, D, A, b, m, S, sig_inv_diag, sig_stride, theta, theta_stride,
               Esde_t, Ef, Edf, dEsde_dm, dEsde_dS, dth_t, dSig_t, status};
    return launch_esde(system, a, vgpa::as_stream(stream));
}

/*
 * The Lorenz-96 unscented energy kernel at D = 40 with S and / or dEsde_dS in the packed symmetric layout
 * (sym_rows.h): `layout` bit 0 (VGPA_PACKED_S): S is (P,N,1152); bit 1 (VGPA_PACKED_DEDS): dEsde_dS is
 * (P,N,1152).  Everything else as vgpa_esde_l96_ut without Edf.
 */
extern "C" int vgpa_esde_l96_ut_packed(int P, int N, int layout, const double* A, const double* b, const double* m,
                                       const double* S, const double* sig_inv_diag, int sig_stride,
                                       const double* theta, int theta_stride, double* Esde_t, double* Ef,
                                       double* dEsde_dm, double* dEsde_dS, double* dth_t, double* dSig_t,
                                       int* status, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1, "vgpa_esde_l96_ut_packed: bad sizes P=%d N=%d", P, N);
    VGPA_REQUIRE(layout >= 0 && layout <= 3, "vgpa_esde_l96_ut_packed: layout %d is not a combination of VGPA_PACKED_S and "
                 "VGPA_PACKED_DEDS", layout);
    VGPA_REQUIRE(A && b && m && S && sig_inv_diag && theta && Esde_t && Ef && dEsde_dm && dEsde_dS,
                 "vgpa_esde_l96_ut_packed: null pointer argument");
    VGPA_REQUIRE((size_t)P * N < 0x7fffffffull, "vgpa_esde_l96_ut_packed: P*N too large");
    EsdeArgs a{P, N, 40, A, b, m, S, sig_inv_diag, sig_stride, theta, theta_stride,
               Esde_t, Ef, nullptr, dEsde_dm, dEsde_dS, dth_t, dSig_t, status};
    a.packed = layout;
    return launch_esde(VGPA_SYS_L96_UT, a, vgpa::as_stream(stream));
}

#define VGPA_DEF_ESDE(name, SYS)                                                                        \
    extern "C" int name(int P, int N, int D, const double* A, const double* b, const double* m,         \
                        const double* S, const double* sig_inv_diag, int sig_stride, const double* theta, \
                        int theta_stride, double* Esde_t, double* Ef, double* Edf, double* dEsde_dm,     \
                        double* dEsde_dS, double* dth_t, double* dSig_t, int* status, void* stream) {    \
        return vgpa_esde(SYS, P, N, D, A, b, m, S, sig_inv_diag, sig_stride, theta, theta_stride, Esde_t, \
                         Ef, Edf, dEsde_dm, dEsde_dS, dth_t, dSig_t, status, stream);                    \
    }
VGPA_DEF_ESDE(vgpa_esde_ou, VGPA_SYS_OU)
VGPA_DEF_ESDE(vgpa_esde_dw, VGPA_SYS_DW)
VGPA_DEF_ESDE(vgpa_esde_l63, VGPA_SYS_L63)
VGPA_DEF_ESDE(vgpa_esde_l63_ut, VGPA_SYS_L63_UT)
VGPA_DEF_ESDE(vgpa_esde_l96_ut, VGPA_SYS_L96_UT)

#ifdef VGPA_STAMPS
extern "C" int vgpa_debug_l96_stamps(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_l96_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
