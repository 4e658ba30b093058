// Relative error of the hardware f64 reciprocal / reciprocal square root estimates and of one
// and two Newton refinements, over random positive inputs (against the host's correctly rounded
// division / sqrt).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void probe(const double* p, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = p[i];
    double q = __builtin_amdgcn_rcp(x);
    out[i] = q;
    double e = fma(-x, q, 1.0);
    q = fma(q, e, q);
    out[n + i] = q;
    e = fma(-x, q, 1.0);
    q = fma(q, e, q);
    out[2 * n + i] = q;
    double r = __builtin_amdgcn_rsq(x);
    out[3 * n + i] = r;
    e = fma(-(x * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    out[4 * n + i] = r;
    e = fma(-(x * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    out[5 * n + i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), o(6 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30.0, 30.0), m(1.0, 2.0);
    for (auto& v : h) v = m(g) * std::exp2(std::floor(u(g)));
    double *d, *r;
    hipMalloc(&d, n * 8); hipMalloc(&r, 6 * n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(d, r, n);
    hipMemcpy(o.data(), r, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"rcp estimate", "rcp + 1 Newton", "rcp + 2 Newton", "rsq estimate", "rsq + 1 Newton", "rsq + 2 Newton"};
    for (int k = 0; k < 6; ++k) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double ref = k < 3 ? 1.0 / h[i] : 1.0 / std::sqrt(h[i]);
            worst = std::fmax(worst, std::fabs(o[k * n + i] - ref) / ref);
        }
        printf("%-16s max relative error %.3e\n", names[k], worst);
    }
    return 0;
}
