// Lorenz-96 unscented energy kernel for D = 40 (esde_l96_wave.h) as a translation unit of its own.
//
// The kernel is 33 KB of straight-line FP64 code bound by instruction issue, and the machine scheduler's
// "max-ilp" strategy makes it 1.5-2 % faster; the same option costs the memory-bound one-lane-per-step
// kernels of esde.hip 30-56 registers, so it is applied to this file alone (vgpa_amd/build.py).
#include "common.h"
#include "esde_args.h"
#include "sym_rows.h"

#include <cmath>
#include <cstdlib>

namespace {

using vgpa::EsdeArgs;
using vgpa::readlane_f64;
using vgpa::wave_lds_fence;

#ifdef VGPA_STAMPS
// Diagnostic build only: per-phase cycle counts of one wavefront.
__device__ unsigned long long g_l96_stamps[32];
#endif

#include "esde_l96_wave.h"

}  // namespace

// The kernel takes a (steps x problems) grid; more problems than a grid dimension holds are launched in
// slices of the problem axis.
void vgpa::launch_esde_l96_wave40(const EsdeArgs& a, hipStream_t st) {
    constexpr int MAXP = 65535;
#ifdef VGPA_EXPERIMENT_LDS_PAD
    // occupancy experiments: a larger LDS request per workgroup lowers the number of resident wavefronts
    static const size_t pad = getenv("VGPA_ESDE_LDS_PAD") ? (size_t)atol(getenv("VGPA_ESDE_LDS_PAD")) : 0;
    const size_t lds = L96Wave::WG_LDS_BYTES + pad;
#else
    const size_t lds = L96Wave::WG_LDS_BYTES;
#endif
    const size_t N = (size_t)a.N, D = 40, DD = 1600;
    const size_t SS = (a.packed & 1) ? vgpa::SymRows40::SIZE : DD, SG = (a.packed & 2) ? vgpa::SymRows40::SIZE : DD;
    for (int p0 = 0; p0 < a.P; p0 += MAXP) {
        EsdeArgs s = a;
        s.P = a.P - p0 < MAXP ? a.P - p0 : MAXP;
        const size_t o = (size_t)p0 * N;
        s.A += o * DD; s.b += o * D; s.m += o * D; s.S += o * SS;
        s.sig_inv_diag += (size_t)p0 * a.sig_stride;
        s.theta += (size_t)p0 * a.theta_stride;
        s.Esde_t += o; s.Ef += o * D; s.dEdm += o * D; s.dEdS += o * SG;
        if (s.Edf) s.Edf += o * DD;
        if (s.dth_t) s.dth_t += o * D;
        if (s.dSig_t) s.dSig_t += o * D;
        if (s.status) s.status += o;
        const dim3 grid((unsigned)((s.N + 1) / 2), (unsigned)s.P);      // two steps per workgroup
        auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(128), lds, st, s); };
        const bool hyper = s.dSig_t != nullptr;
        switch (s.packed & 3) {
            case 3: hyper ? launch(esde_l96_ut_wave40_kernel<true, true, true>) : launch(esde_l96_ut_wave40_kernel<false, true, true>); break;
            case 2: hyper ? launch(esde_l96_ut_wave40_kernel<true, false, true>) : launch(esde_l96_ut_wave40_kernel<false, false, true>); break;
            case 1: hyper ? launch(esde_l96_ut_wave40_kernel<true, true, false>) : launch(esde_l96_ut_wave40_kernel<false, true, false>); break;
            default: hyper ? launch(esde_l96_ut_wave40_kernel<true, false, false>) : launch(esde_l96_ut_wave40_kernel<false, false, false>); break;
        }
    }
}

#ifdef VGPA_STAMPS
extern "C" int vgpa_debug_l96_wave_stamps(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_l96_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
