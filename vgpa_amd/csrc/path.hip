// Input preparation on the device: sample paths of the stochastic systems and the starting
// values of the variational parameters, for P independent problems at once.
//
// vgpa_sde_path   the Euler-Maruyama recursions of `system_path` in dynamics/sys_*.py:
//                 an optional deterministic burn-in from x0 with step dtau
//                 (sys_lorenz_96_ut.py:82-84, sys_lorenz_63.py:38-40), then
//                 X[t] = X[t-1] + f(X[t-1]) dt + K e[t]   (sys_lorenz_96_ut.py:105-106,
//                 sys_lorenz_63.py:64-65, sys_double_well.py:71-72, sys_ornstein_uhlenbeck.py:57-58)
//                 with the random numbers e supplied by the caller (the host modules draw
//                 them from NumPy exactly as the reference does, so a seeded run reproduces
//                 the reference's path bit for bit).
// vgpa_init_ab0   the n-dimensional branch of auxiliary/initialize_Ab0.py:66-93 once the
//                 spline means are known.
//
// A path is sequential in time; one wavefront walks one problem, lane i owns components
// i, i + 64, ...; the state sits in two alternating LDS buffers (the Lorenz-96 drift couples
// a component to three neighbours on the ring).  Products and sums round separately, in the
// order the reference's NumPy expressions evaluate them.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct PathArgs {
    int system, N, D;
    const double* x0;
    const double* add0;
    const double* theta;
    int theta_stride;
    const double* kdiag;
    int k_stride;
    int burn;
    double dtau, dt;
    const double* noise;
    double* X;
};

// drift of component i at state x (D components in LDS)
__device__ inline double path_drift(int system, int D, int i, const double* x, const double* th) {
    switch (system) {
        case VGPA_SYS_OU:                                   // thet0*(mu - x), mu = 0
            return th[0] * (0.0 - x[0]);
        case VGPA_SYS_DW:                                   // 4*x*(thet0 - x**2)
            return (4.0 * x[0]) * (th[0] - x[0] * x[0]);
        case VGPA_SYS_L63:
        case VGPA_SYS_L63_UT:                               // sys_lorenz_63.py:32-34
            if (i == 0) return (-th[0]) * (x[0] - x[1]);
            if (i == 1) return (th[1] - x[2]) * x[0] - x[1];
            return x[0] * x[1] - th[2] * x[2];
        default: {                                          // (f1 - b2)*b1 - x + f, sys_lorenz_96_ut.py:36-38
            const int f1 = (i + 1 == D) ? 0 : i + 1, b1 = (i == 0) ? D - 1 : i - 1, b2 = (i <= 1) ? i + D - 2 : i - 2;
            return (x[f1] - x[b2]) * x[b1] - x[i] + th[0];
        }
    }
}

__device__ inline void path_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

__global__ void __launch_bounds__(64) sde_path_kernel(PathArgs a) {
    extern __shared__ double lds[];
    const int D = a.D, lane = threadIdx.x;
    const size_t p = blockIdx.x;
    double* cur = lds;
    double* nxt = lds + D;
    double th[3];
    const int nth = (a.system == VGPA_SYS_L63 || a.system == VGPA_SYS_L63_UT) ? 3 : 1;
    for (int k = 0; k < 3; ++k) th[k] = k < nth ? a.theta[p * a.theta_stride + k] : 0.0;
    for (int i = lane; i < D; i += 64) cur[i] = a.x0[p * D + i];
    path_fence();
    for (int s = 0; s < a.burn; ++s) {
        for (int i = lane; i < D; i += 64) nxt[i] = cur[i] + path_drift(a.system, D, i, cur, th) * a.dtau;
        path_fence();
        double* t = cur; cur = nxt; nxt = t;
    }
    double* Xp = a.X + p * (size_t)a.N * D;
    const double* Ep = a.noise + p * (size_t)a.N * D;
    if (a.add0) {
        for (int i = lane; i < D; i += 64) nxt[i] = cur[i] + a.add0[p * D + i];
        path_fence();
        double* t = cur; cur = nxt; nxt = t;
    }
    for (int i = lane; i < D; i += 64) Xp[i] = cur[i];
    for (int t = 1; t < a.N; ++t) {
        for (int i = lane; i < D; i += 64) {
            const double k = a.kdiag[p * a.k_stride + i];
            const double v = (cur[i] + path_drift(a.system, D, i, cur, th) * a.dt) + Ep[(size_t)t * D + i] * k;
            nxt[i] = v;
            Xp[(size_t)t * D + i] = v;
        }
        path_fence();
        double* q = cur; cur = nxt; nxt = q;
    }
}

// A0[k] = 0.5 diag(Sigma_ii / S0_ii);  b0[k] = (mt[k+1] - mt[k]) / dt + A_ii mt[k],  b0[N-1] = A_ii mt[N-1]
__global__ void init_ab0_kernel(size_t total, int N, int D, const double* __restrict__ mt, const double* __restrict__ sig,
                                int sig_stride, const double* __restrict__ s0, int s0_stride, double dt,
                                double* __restrict__ A0, double* __restrict__ b0) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (p, k, i, j)
    if (q >= total) return;
    const int j = (int)(q % D), i = (int)((q / D) % D);
    const size_t pk = q / ((size_t)D * D);
    const int k = (int)(pk % N);
    const size_t p = pk / N;
    double v = 0.0;
    if (i == j) {
        const double aii = 0.5 * (sig[p * sig_stride + i] / s0[p * s0_stride + i]);
        v = aii;
        const double mk = mt[pk * D + i];
        b0[pk * D + i] = (k + 1 < N) ? (mt[(pk + 1) * D + i] - mk) / dt + aii * mk : aii * mk;
    }
    A0[q] = v;
}

}  // namespace

extern "C" int vgpa_sde_path(int system, int P, int N, int D, const double* x0, const double* add0,
                             const double* theta, int theta_stride, const double* kdiag, int k_stride, int burn,
                             double dtau, double dt, const double* noise, double* X, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 1 && burn >= 0, "vgpa_sde_path: bad sizes P=%d N=%d D=%d burn=%d", P, N, D, burn);
    VGPA_REQUIRE(x0 && theta && kdiag && noise && X, "vgpa_sde_path: null pointer argument");
    VGPA_REQUIRE(k_stride == D || k_stride == 0, "vgpa_sde_path: k_stride must be 0 (one (D) vector for all problems) or D");
    switch (system) {
        case VGPA_SYS_OU:
        case VGPA_SYS_DW:
            VGPA_REQUIRE(D == 1, "vgpa_sde_path: system %d needs D == 1 (got %d)", system, D);
            break;
        case VGPA_SYS_L63:
        case VGPA_SYS_L63_UT:
            VGPA_REQUIRE(D == 3, "vgpa_sde_path: Lorenz-63 needs D == 3 (got %d)", D);
            break;
        case VGPA_SYS_L96_UT:
            VGPA_REQUIRE(D >= 4, "vgpa_sde_path: Lorenz-96 needs D >= 4 (got %d)", D);
            break;
        default:
            return vgpa::fail(VGPA_EINVAL, "vgpa_sde_path: unknown system %d", system);
    }
    VGPA_REQUIRE(D <= 4096, "vgpa_sde_path: D=%d too large (at most 4096 components)", D);
    if (P == 0) return VGPA_OK;
    PathArgs a{system, N, D, x0, add0, theta, theta_stride, kdiag, k_stride, burn, dtau, dt, noise, X};
    hipLaunchKernelGGL(sde_path_kernel, dim3((unsigned)P), dim3(64), sizeof(double) * 2 * D, vgpa::as_stream(stream), a);
    return vgpa::check_launch("vgpa_sde_path");
}

extern "C" int vgpa_init_ab0(int P, int N, int D, const double* mt0, const double* sig_diag, int sig_stride,
                             const double* s0_diag, int s0_stride, double dt, double* A0, double* b0, void* stream) {
    VGPA_REQUIRE(P >= 0 && N >= 1 && D >= 2, "vgpa_init_ab0: bad sizes P=%d N=%d D=%d (the 1-D case is a host formula)", P, N, D);
    VGPA_REQUIRE(mt0 && sig_diag && s0_diag && A0 && b0, "vgpa_init_ab0: null pointer argument");
    VGPA_REQUIRE(dt != 0.0, "vgpa_init_ab0: dt must not be zero");
    const size_t total = (size_t)P * N * D * D;
    if (total == 0) return VGPA_OK;
    VGPA_REQUIRE((total + 255) / 256 < 0x7fffffffull, "vgpa_init_ab0: P*N*D*D too large");
    hipLaunchKernelGGL(init_ab0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vgpa::as_stream(stream), total, N,
                       D, mt0, sig_diag, sig_stride, s0_diag, s0_stride, dt, A0, b0);
    return vgpa::check_launch("vgpa_init_ab0");
}
