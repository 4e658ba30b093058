// Arguments of the per-time-step energy kernels, shared by the translation units that hold them.
#pragma once

#include <hip/hip_runtime.h>

namespace vgpa {

struct EsdeArgs {
    int P, N, D;
    const double *A, *b, *m, *S;
    const double* sig_inv_diag;
    int sig_stride;
    const double* theta;
    int theta_stride;
    double *Esde_t, *Ef, *Edf, *dEdm, *dEdS, *dth_t, *dSig_t;
    int* status;
    int packed = 0;     // Lorenz-96, D = 40 only, layout of sym_rows.h: bit 0 = S is packed, bit 1 = dEdS is packed
};

// Lorenz-96, D = 40: the wavefront-per-step kernel (esde_l96_wave.hip, a translation unit of its own
// because it is compiled with another instruction-scheduling strategy than the memory-bound kernels).
void launch_esde_l96_wave40(const EsdeArgs& a, hipStream_t st);

// Cross-lane read of a double from a lane named by a scalar.
__device__ inline double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Orders LDS traffic between the lanes of one wavefront (no instruction is
// emitted; it only stops the compiler from moving LDS accesses across it).
__device__ inline void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

}  // namespace vgpa
