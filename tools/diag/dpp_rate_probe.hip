// Issue cost (cycles per instruction, per wavefront) of the f64 instructions the Lorenz-96
// wavefront kernel uses in its pivot loop, with 1 and 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(64) rate(double* out, long long* cyc) {
    double x[8], t = out[threadIdx.x], xk = out[64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = out[128 + i * 64 + threadIdx.x];
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 200; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(t), "v"(xk));
            if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(t), "v"(xk));
            if (MODE == 2) { double b; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_e32 %2, %0, %3" : "=&v"(b), "+v"(t), "+v"(x[i]) : "v"(xk)); }
            if (MODE == 3) asm volatile("v_add_f64 %0, %1, %2" : "+v"(x[i]) : "v"(t), "v"(xk));
            if (MODE == 4) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(t), i), hi = __builtin_amdgcn_readlane(__double2hiint(t), i);
                x[i] = fma(__hiloint2double(hi, lo), xk, x[i]);
                asm volatile("" : "+v"(x[i]));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* d; long long* c; hipMalloc(&d, 64 * 16 * 8); hipMalloc(&c, 8); hipMemset(d, 0, 64 * 16 * 8);
    const char* names[5] = {"v_fmac_f64", "v_fmac_f64_dpp", "v_mov_b64_dpp+v_fmac_f64", "v_add_f64", "2 x v_readlane + v_fma_f64"};
    for (int blocks : {1, 256 * 8, 256 * 16}) {   // 1 wave alone; 2 waves/SIMD; 4 waves/SIMD (64-thread blocks)
        for (int m = 0; m < 5; ++m) {
            long long h = 0;
            if (m == 0) rate<0><<<blocks, 64>>>(d, c);
            if (m == 1) rate<1><<<blocks, 64>>>(d, c);
            if (m == 2) rate<2><<<blocks, 64>>>(d, c);
            if (m == 3) rate<3><<<blocks, 64>>>(d, c);
            if (m == 4) rate<4><<<blocks, 64>>>(d, c);
            hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            printf("blocks %5d  %-28s %.2f ticks per instruction\n", blocks, names[m], h / 1600.0);
        }
    }
    return 0;
}
