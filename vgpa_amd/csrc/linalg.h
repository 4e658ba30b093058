// Small dense linear algebra shared by the kernels: one wavefront (64 threads, the
// whole block) working on matrices in LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace vgpa {

// In-place Gauss-Jordan inverse with partial pivoting of the D x D matrix M
// (row stride LD) using the D x D scratch I (same stride); result in I.
__device__ inline void gj_inverse(double* M, double* I, int D, int LD, int l) {
    for (int q = l; q < D * D; q += 64) I[(q / D) * LD + (q % D)] = (q / D == q % D) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = 0; k < D; ++k) {
        // pivot search over rows k..D-1 of column k
        double best = -1.0;
        int brow = k;
        for (int r = k + l; r < D; r += 64) {
            const double v = fabs(M[r * LD + k]);
            if (v > best) { best = v; brow = r; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int orow = __shfl_xor(brow, off, 64);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
        }
        if (brow != k) {
            for (int j = l; j < D; j += 64) {
                double t = M[k * LD + j]; M[k * LD + j] = M[brow * LD + j]; M[brow * LD + j] = t;
                t = I[k * LD + j]; I[k * LD + j] = I[brow * LD + j]; I[brow * LD + j] = t;
            }
        }
        __syncthreads();
        const double rp = 1.0 / M[k * LD + k];
        __syncthreads();
        for (int j = l; j < D; j += 64) { M[k * LD + j] *= rp; I[k * LD + j] *= rp; }
        __syncthreads();
        for (int r = 0; r < D; ++r) {
            if (r == k) continue;
            const double f = M[r * LD + k];
            __syncthreads();
            for (int j = l; j < D; j += 64) {
                M[r * LD + j] = fma(-f, M[k * LD + j], M[r * LD + j]);
                I[r * LD + j] = fma(-f, I[k * LD + j], I[r * LD + j]);
            }
        }
        __syncthreads();
    }
}

}  // namespace vgpa
