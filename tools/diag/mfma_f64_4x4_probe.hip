// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: lane layout of the four independent 4x4x4 blocks and the
// issue rate, beside v_mfma_f64_16x16x4_f64 (tools/diag/mfma_f64_probe.hip).  Question behind it: would
// 4x4 blocks, which tile D = 40 without padding, issue the multiply-adds faster than 16x16 tiles of the
// matrices padded to 48?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// The three 2-bit fields of a lane number, lowest first; perm p (0..5) says which field is which index.
__host__ __device__ inline void fields(int l, int p, int& f0, int& f1, int& f2) {
    const int v[3] = {l & 3, (l >> 2) & 3, (l >> 4) & 3};
    const int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    f0 = v[P[p][0]]; f1 = v[P[p][1]]; f2 = v[P[p][2]];
}

__global__ void layout(const double* A, const double* B, double* C, int pa, int pb) {
    // A operand of lane l = A[blk][i][k] with (blk, i, k) the fields of l in the order perm pa names;
    // B operand = B[blk][k][j] with (blk, k, j) in the order perm pb names
    const int l = threadIdx.x;
    int blk, i, k, j;
    fields(l, pa, blk, i, k);
    const double a = A[blk * 16 + i * 4 + k];
    fields(l, pb, blk, k, j);
    const double b = B[blk * 16 + k * 4 + j];
    C[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

__global__ void rate(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void rate_dep(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3, c0 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    std::vector<double> A(64), B(64), C(64), R(64);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i; B[i] = 100 + 3 * i + (i * i) % 7; }
    for (int blk = 0; blk < 4; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += A[blk * 16 + i * 4 + k] * B[blk * 16 + k * 4 + j];
                R[blk * 16 + i * 4 + j] = s;
            }
    double *dA, *dB, *dC; long long* dcyc;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, (size_t)2048 * 1024 * 8); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    // try every assignment of the lane's three 2-bit fields to (block, row, k) / (block, k, column) / (block, row, column)
    const char* names[6] = {"(f0,f1,f2)", "(f0,f2,f1)", "(f1,f0,f2)", "(f1,f2,f0)", "(f2,f0,f1)", "(f2,f1,f0)"};
    int found = 0;
    for (int pa = 0; pa < 6; ++pa)
        for (int pb = 0; pb < 6; ++pb) {
            layout<<<1, 64>>>(dA, dB, dC, pa, pb);
            hipMemcpy(C.data(), dC, 64 * 8, hipMemcpyDeviceToHost);
            for (int pc = 0; pc < 6; ++pc) {
                int bad = 0;
                for (int l = 0; l < 64; ++l) {
                    int blk, i, j;
                    fields(l, pc, blk, i, j);
                    bad += C[l] != R[blk * 16 + i * 4 + j];
                }
                if (bad == 0) {
                    printf("layout (lane fields f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4): A (blk,i,k) = %s, B (blk,k,j) = %s, result (blk,i,j) = %s\n",
                           names[pa], names[pb], names[pc]);
                    ++found;
                }
            }
        }
    if (!found) printf("layout: none of the 216 field assignments reproduces the four block products\n");
    const int iters = 10000; long long cyc;
    for (int waves = 1; waves <= 8; waves *= 2) {
        rate<<<1, 64 * waves>>>(dC, dcyc, iters); hipDeviceSynchronize();
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        printf("mfma f64 4x4x4 (4 blocks, 256 multiply-adds), %d wave(s)/block, 8 independent acc: %.1f cycles per instruction per wave\n",
               waves, (double)cyc / (8.0 * iters));
    }
    rate_dep<<<1, 64>>>(dC, dcyc, iters); hipDeviceSynchronize();
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    printf("mfma f64 4x4x4 dependent chain: %.1f cycles per instruction\n", (double)cyc / (4.0 * iters));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {4, 8, 16}) {
        rate<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipDeviceSynchronize();
        hipEventRecord(e0); rate<<<2048, 64 * wpb>>>(dC, dcyc, 2000); hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        double flop = 2048.0 * wpb * 2000.0 * 8 * 512.0;   // 256 multiply-adds = 512 flop per instruction
        printf("chip-wide mfma f64 4x4x4, %d waves/block: %.1f TFLOP/s (%.3f ms)\n", wpb, flop / ms / 1e9, ms);
    }
    return 0;
}
