// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane layout and issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout(const double* A, const double* B, double* C) {
    // A 16x4 row-major, B 4x16 row-major, C 16x16 row-major
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void rate(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void rate_dep(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double4_t c0 = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void rate_fma(double* out, long long* cyc, int iters) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = fma(a, b, c[q]);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += c[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    std::vector<double> A(64), B(64), C(256), R(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 4 + k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 100 + 7 * k + 3 * j + (k * j) % 5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    double *dA, *dB, *dC; long long* dcyc;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, (size_t)2048 * 1024 * 8 * 2); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
    printf("layout mismatches: %d of 256\n", bad);
    const int iters = 10000; long long cyc;
    for (int waves = 1; waves <= 16; waves *= 2) {
        rate<<<1, 64 * waves>>>(dC, dcyc, iters); hipDeviceSynchronize();
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        printf("mfma f64 16x16x4, %d wave(s)/block, 4 independent acc: %.1f cycles per mfma per wave\n", waves, (double)cyc / (4.0 * iters));
    }
    {   // absolute rate over the whole chip
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int wpb : {4, 8, 16}) {
            rate<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipDeviceSynchronize();
            hipEventRecord(e0); rate<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipEventRecord(e1); hipDeviceSynchronize();
            float ms = 0; hipError_t er = hipEventElapsedTime(&ms, e0, e1);
            if (er != hipSuccess) printf("event error %s / last %s\n", hipGetErrorString(er), hipGetErrorString(hipGetLastError()));
            double flop = 2048.0 * wpb * 2000.0 * 4 * 2048.0;   // 1024 FMA = 2048 flop per MFMA
            printf("chip-wide mfma f64, %d waves/block: %.1f TFLOP/s (%.3f ms)\n", wpb, flop / ms / 1e9, ms);
        }
        for (int wpb : {4, 8, 16}) {
            rate_fma<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipDeviceSynchronize();
            hipEventRecord(e0); rate_fma<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = 2048.0 * wpb * 64 * 2000.0 * 8 * 2.0;
            printf("chip-wide v_fma_f64, %d waves/block: %.1f TFLOP/s (%.3f ms)\n", wpb, flop / ms / 1e9, ms);
        }
    }
    rate_dep<<<1, 64>>>(dC, dcyc, iters); hipDeviceSynchronize();
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    printf("mfma f64 dependent chain: %.1f cycles per mfma\n", (double)cyc / (4.0 * iters));
    for (int waves = 1; waves <= 8; waves *= 2) {
        rate_fma<<<1, 64 * waves>>>(dC, dcyc, iters); hipDeviceSynchronize();
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64, %d wave(s)/block, 8 independent: %.2f cycles per fma per wave\n", waves, (double)cyc / (8.0 * iters));
    }
    return 0;
}
