// What HBM delivers on this part for the access mixes of the D = 40 kernels: grid-stride kernels
// with sixteen bytes per lane, R read streams and W write streams of 1 GiB each.
//   1R      read only              1R1W   copy
//   1R1Wx   the forward scan's mix (reads 12.8 KB + 0.3 KB, writes the same)
//   3R2W    the fused backward scan + gradient's mix (three matrices in, two out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int W>
__global__ void __launch_bounds__(256) stream(const double2* const* in, double2* const* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double2 acc = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 v = in[r][i];
            acc.x += v.x;
            acc.y += v.y;
        }
        if (W == 0) {
            if (acc.x == 12345.678) out[0][0] = acc;
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) out[w][i] = acc;
        }
    }
}

template <int R, int W>
void run(const char* name, double2** d_in, double2** d_out, size_t n, int blocks) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    stream<R, W><<<blocks, 256>>>(d_in, d_out, n);
    (void)hipEventRecord(a);
    for (int k = 0; k < 5; ++k) stream<R, W><<<blocks, 256>>>(d_in, d_out, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 5.0 * (R + W) * n * sizeof(double2);
    printf("%-6s %6d workgroups: %7.1f GB/s\n", name, blocks, bytes / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t n = (size_t)1 << 26;   // 2^26 x 16 B = 1 GiB per stream
    std::vector<double2*> in(3), out(2);
    for (auto& p : in) { (void)hipMalloc(&p, n * sizeof(double2)); (void)hipMemset(p, 0, n * sizeof(double2)); }
    for (auto& p : out) { (void)hipMalloc(&p, n * sizeof(double2)); (void)hipMemset(p, 0, n * sizeof(double2)); }
    double2 **d_in, **d_out;
    (void)hipMalloc(&d_in, 3 * sizeof(double2*));
    (void)hipMalloc(&d_out, 2 * sizeof(double2*));
    (void)hipMemcpy(d_in, in.data(), 3 * sizeof(double2*), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_out, out.data(), 2 * sizeof(double2*), hipMemcpyHostToDevice);
    for (int blocks : {1024, 4096, 16384}) {
        run<1, 0>("1R", d_in, d_out, n, blocks);
        run<1, 1>("1R1W", d_in, d_out, n, blocks);
        run<2, 1>("2R1W", d_in, d_out, n, blocks);
        run<3, 2>("3R2W", d_in, d_out, n, blocks);
    }
    return 0;
}
