// Do FP64 matrix-core instructions overlap with vector-ALU work of another wavefront on the
// same SIMD?  A 512-thread workgroup puts wavefronts w and w + 4 on one SIMD.  Wavefronts 0..3
// run role A, wavefronts 4..7 role B; each role is timed alone and together with the other.
//   roles: 0 idle, 1 v_mfma_f64_16x16x4_f64 (4 accumulators), 2 v_fma_f64 (16 chains),
//          3 32-bit integer v_mad (16 chains), 4 v_fma_f32 (16 chains), 5 v_mfma_f64 on one accumulator
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ long long run_role(int iters, double* sink, int tid) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (ROLE == 1) {
        f64x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const double a = 1.0 + tid, b = 0.5;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
        if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.0) sink[0] = 1.0;
    } else if (ROLE == 5) {
        // one accumulator: every instruction depends on the previous one
        f64x4 c = {0, 0, 0, 0};
        const double a = 1.0 + tid, b = 0.5;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        if (c[0] + c[1] == 12345.0) sink[0] = 1.0;
    } else if (ROLE == 2) {
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = tid + k;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = fma(x[k], 1.0000001, 0.5);
        double s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += x[k];
        if (s == 12345.0) sink[0] = 1.0;
    } else if (ROLE == 3) {
        int x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = tid + k;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = x[k] * 3 + (x[k] >> 1);
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += x[k];
        if (s == 12345) sink[0] = 1.0;
    } else if (ROLE == 4) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = tid + k;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = fmaf(x[k], 1.0000001f, 0.5f);
        float s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += x[k];
        if (s == 12345.0f) sink[0] = 1.0;
    }
    return __builtin_amdgcn_s_memtime() - t0;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512) probe(int iters, long long* out, double* sink) {
    const int w = threadIdx.x >> 6;
    long long t;
    if (w < 4) t = run_role<RA>(iters, sink, threadIdx.x);
    else t = run_role<RB>(iters, sink, threadIdx.x);
    if ((threadIdx.x & 63) == 0) out[w] = t;
}

template <int RA, int RB>
void go(const char* name, int iters, double per_a, double per_b) {
    long long* d; double* s;
    (void)hipMalloc(&d, 8 * sizeof(long long));
    (void)hipMalloc(&s, 8);
    probe<RA, RB><<<1, 512>>>(iters, d, s);
    probe<RA, RB><<<1, 512>>>(iters, d, s);
    long long h[8];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on this part: print ticks and the ratio only
    printf("%-34s A (wavefront 0): %8lld ticks = %6.2f per instr   B (wavefront 4): %8lld ticks = %6.2f per instr\n", name, h[0],
           per_a > 0 ? h[0] / (iters * per_a) : 0.0, h[4], per_b > 0 ? h[4] / (iters * per_b) : 0.0);
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    const int it = 20000;
    go<1, 0>("mfma f64 alone", it, 4, 0);
    go<0, 2>("fma f64 alone", it, 0, 16);
    go<1, 2>("mfma f64 + fma f64", it, 4, 16);
    go<0, 3>("int mad alone", it, 0, 32);
    go<1, 3>("mfma f64 + int", it, 4, 32);
    go<2, 3>("fma f64 + int", it, 16, 32);
    go<0, 4>("fma f32 alone", it, 0, 16);
    go<1, 4>("mfma f64 + fma f32", it, 4, 16);
    go<2, 2>("fma f64 + fma f64", it, 16, 16);
    go<1, 1>("mfma f64 + mfma f64", it, 4, 4);
    go<5, 0>("mfma f64, one accumulator", it, 4, 0);
    go<5, 2>("mfma f64 (one acc.) + fma f64", it, 4, 16);
    return 0;
}
