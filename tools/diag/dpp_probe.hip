// Semantics of the gfx950 cross-lane instructions the Lorenz-96 wavefront kernel relies on:
// v_mov_b64_dpp / v_fmac_f64_dpp with row_newbcast, and v_permlane16_swap_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    double a = 100.0 + l;            // value identifying the lane
    double d;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a));
    double c = 0.0, one = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(one));
    int x = l, y = 1000 + l;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[l] = d;
    out[64 + l] = c;
    out[128 + l] = r[0];
    out[192 + l] = r[1];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    probe<<<1, 64>>>(d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"mov_b64 row_newbcast:5", "fmac row_newbcast:3 (src0 * 1)", "permlane16_swap result[0] (vdst=lane)", "permlane16_swap result[1] (src=1000+lane)"};
    for (int k = 0; k < 4; ++k) {
        printf("%s\n", names[k]);
        for (int l = 0; l < 64; ++l) printf("%6.0f%s", h[64 * k + l], (l % 16 == 15) ? "\n" : "");
    }
    return 0;
}
