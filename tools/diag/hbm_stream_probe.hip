// What HBM delivers on this part for the access mixes of the D = 40 kernels.
//
// Streaming kernels with R read streams and W write streams of 1 GiB each, 16 bytes per lane and
// access, pointers as __restrict__ kernel arguments, U independent loads of every stream in flight
// per lane before the first use (U x R x 16 bytes per lane), a workgroup sweeping a contiguous
// block of the streams.  Each mix is run with plain accesses ("plain"), with non-temporal stores
// ("ntst") and with non-temporal loads and stores ("nt"), for several grid sizes and unroll depths;
// the line format is   <mix> <variant> U=<u> <workgroups> workgroups: <GB/s> GB/s
//   1R      read only              1R1W   copy (forward scan: A in, S out)
//   2R1W    the energy kernel's mix (A, S in; dEsde/dS out)
//   3R2W    the fused backward scan + gradient's mix (A, S, dEsde/dS in; Psi, dL/dA out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ v2d ld(const v2d* p) {
    if (MODE == 2) return __builtin_nontemporal_load(p);
    return *p;
}
template <int MODE>
__device__ __forceinline__ void st(v2d* p, v2d v) {
    if (MODE >= 1) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0 plain, 1 non-temporal stores, 2 non-temporal loads and stores
template <int R, int W, int U, int MODE>
__global__ void __launch_bounds__(256)
stream(const v2d* __restrict__ i0, const v2d* __restrict__ i1, const v2d* __restrict__ i2, v2d* __restrict__ o0,
       v2d* __restrict__ o1, size_t n) {
    const v2d* in[3] = {i0, i1, i2};
    v2d* out[2] = {o0, o1};
    // a workgroup takes tiles of 256 * U consecutive 16-byte pieces, grid-stride over tiles
    const size_t tile = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
        v2d v[R][U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) v[r][u] = ld<MODE>(in[r] + base + (size_t)u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2d acc = v[0][u];
#pragma unroll
            for (int r = 1; r < R; ++r) acc += v[r][u];
            if (W == 0) {
                if (acc.x == 12345.678) *o0 = acc;
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) st<MODE>(out[w] + base + (size_t)u * 256 + threadIdx.x, acc);
            }
        }
    }
}

template <int R, int W, int U, int MODE>
void run(const char* name, v2d** in, v2d** out, size_t n, int blocks) {
    static const char* variant[3] = {"plain", "ntst", "nt"};
    if (W == 0 && MODE == 1) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    stream<R, W, U, MODE><<<blocks, 256>>>(in[0], in[1], in[2], out[0], out[1], n);
    (void)hipEventRecord(a);
    for (int k = 0; k < 5; ++k) stream<R, W, U, MODE><<<blocks, 256>>>(in[0], in[1], in[2], out[0], out[1], n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = 5.0 * (R + W) * n * sizeof(v2d);
    printf("%-5s %-5s U=%d %6d workgroups: %7.1f GB/s\n", name, variant[MODE], U, blocks, bytes / (ms * 1e-3) / 1e9);
    fflush(stdout);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
}

template <int R, int W, int U>
void modes(const char* name, v2d** in, v2d** out, size_t n, int blocks) {
    run<R, W, U, 0>(name, in, out, n, blocks);
    run<R, W, U, 1>(name, in, out, n, blocks);
    run<R, W, U, 2>(name, in, out, n, blocks);
}

int main() {
    const size_t n = (size_t)1 << 26;   // 2^26 x 16 B = 1 GiB per stream
    std::vector<v2d*> in(3), out(2);
    for (auto& p : in) {
        if (hipMalloc(&p, n * sizeof(v2d)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        (void)hipMemset(p, 0, n * sizeof(v2d));
    }
    for (auto& p : out) {
        if (hipMalloc(&p, n * sizeof(v2d)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        (void)hipMemset(p, 0, n * sizeof(v2d));
    }
    // 256 compute units: 2048 / 4096 workgroups of four wavefronts = 8 / 16 resident wavefronts per CU and
    // more (the grid-stride loop makes them persistent), 16384 = many short-lived ones
    for (int blocks : {2048, 4096, 16384}) {
        modes<1, 0, 4>("1R", in.data(), out.data(), n, blocks);
        modes<1, 0, 8>("1R", in.data(), out.data(), n, blocks);
        modes<1, 1, 4>("1R1W", in.data(), out.data(), n, blocks);
        modes<1, 1, 8>("1R1W", in.data(), out.data(), n, blocks);
        modes<2, 1, 2>("2R1W", in.data(), out.data(), n, blocks);
        modes<2, 1, 4>("2R1W", in.data(), out.data(), n, blocks);
        modes<3, 2, 2>("3R2W", in.data(), out.data(), n, blocks);
        modes<3, 2, 4>("3R2W", in.data(), out.data(), n, blocks);
    }
    return 0;
}
