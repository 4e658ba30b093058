// How fast is a one-wavefront-per-matrix 40x40 Cholesky when many wavefronts share a CU?
// Lane i owns row i in registers; column j is broadcast through a small LDS buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
constexpr int D = 40;

__device__ inline double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ inline double rsqrt_refined(double p) {
    double r = __builtin_amdgcn_rsq(p);
    double e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-(p * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(64) chol_kernel(const double* __restrict__ S, double* __restrict__ L, size_t total) {
    __shared__ double cb[2][64];
    const int lane = threadIdx.x;
    const int i = min(lane, D - 1);
    for (size_t e = blockIdx.x; e < total; e += gridDim.x) {
        const double* Se = S + e * D * D;
        double r[D];
        double keep = 0.0;
#pragma unroll 1
        for (int rep = 0; rep < (MODE == 2 ? 4 : 1); ++rep) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < D; c += 2) {
            const double2 v = *reinterpret_cast<const double2*>(Se + i * D + c);
            r[c] = v.x; r[c + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < (MODE == 1 ? 0 : D); ++j) {
            const double piv = readlane_f64(r[j], j);
            const double ri = rsqrt_refined(piv);
            r[j] *= ri;
            if (j + 1 < D) {
                const double lnext = readlane_f64(r[j], j + 1);
                r[j + 1] = fma(-r[j], lnext, r[j + 1]);
            }
            if (j + 2 < D) {
                double* c = cb[j & 1];
                c[lane] = r[j];
                int k = j + 2;
                if (k & 1) { r[k] = fma(-r[j], c[k], r[k]); ++k; }
#pragma unroll
                for (; k < D; k += 2) {
                    const double2 l = *reinterpret_cast<const double2*>(c + k);
                    r[k] = fma(-r[j], l.x, r[k]);
                    r[k + 1] = fma(-r[j], l.y, r[k + 1]);
                }
            }
        }
        keep += r[D - 1];
        }
        if (MODE == 2 && keep == 12345.0) r[0] = keep;
        if (lane < D) {
            double* Le = L + e * D * D;
#pragma unroll
            for (int c = 0; c < D; c += 2)
                *reinterpret_cast<double2*>(Le + i * D + c) = make_double2(c <= i ? r[c] : 0.0, c + 1 <= i ? r[c + 1] : 0.0);
        }
    }
}

int main(int argc, char** argv) {
    const size_t total = 256 * 2002;
    std::vector<double> h(D * D);
    for (int a = 0; a < D; ++a) for (int b = 0; b < D; ++b) h[a * D + b] = (a == b ? 2.0 : 0.0) + 0.3 / (1 + abs(a - b));
    double *S, *L;
    (void)hipMalloc(&S, total * D * D * 8); (void)hipMalloc(&L, total * D * D * 8);
    for (size_t e = 0; e < total; e += 4096) {}
    // replicate the matrix
    std::vector<double> big(1024 * D * D);
    for (int e = 0; e < 1024; ++e) for (int q = 0; q < D * D; ++q) big[e * D * D + q] = h[q] * (1.0 + 1e-3 * e);
    for (size_t e = 0; e < total; e += 1024) (void)hipMemcpy(S + e * D * D, big.data(), std::min<size_t>(1024, total - e) * D * D * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {4, 8, 12, 16}) {
        const int grid = 256 * wpc;
        chol_kernel<0><<<grid, 64>>>(S, L, total); (void)hipDeviceSynchronize();
        hipEventRecord(e0); chol_kernel<0><<<grid, 64>>>(S, L, total); hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d (%2d waves per CU): %.3f ms for %zu matrices = %.0f cycles per matrix per CU at 2.4 GHz; err %s\n", grid, wpc, ms, total,
               ms * 1e-3 * 2.4e9 * 256 / total, hipGetErrorString(hipGetLastError()));
    }
    for (int wpc : {8, 16}) {
        const int grid = 256 * wpc;
        chol_kernel<1><<<grid, 64>>>(S, L, total); (void)hipDeviceSynchronize();
        hipEventRecord(e0); chol_kernel<1><<<grid, 64>>>(S, L, total); hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("load + store only, %2d waves per CU: %.3f ms\n", wpc, ms);
    }
    for (int wpc : {4, 8, 12, 16}) {
        const int grid = 256 * wpc;
        chol_kernel<2><<<grid, 64>>>(S, L, total); (void)hipDeviceSynchronize();
        hipEventRecord(e0); chol_kernel<2><<<grid, 64>>>(S, L, total); hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("4 factorisations per load, %2d waves per CU: %.3f ms -> (t - 3.6)/3 = %.3f ms per pass = %.0f cycles per matrix per CU\n", wpc, ms, (ms - 3.6) / 3, (ms - 3.6) / 3 * 1e-3 * 2.4e9 * 256 / total);
    }
    chol_kernel<0><<<4096, 64>>>(S, L, total); (void)hipDeviceSynchronize();
    // check one factor
    std::vector<double> l(D * D);
    (void)hipMemcpy(l.data(), L, D * D * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int a = 0; a < D; ++a) for (int b = 0; b <= a; ++b) { double s = 0; for (int k = 0; k < D; ++k) s += l[a * D + k] * l[b * D + k]; worst = fmax(worst, fabs(s - h[a * D + b])); }
    printf("max |L L' - S| = %.3e\n", worst);
    return 0;
}
