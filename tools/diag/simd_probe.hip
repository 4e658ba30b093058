// Where do the wavefronts of a 256-thread workgroup land?  Prints the SIMD ids of
// the four wavefronts for a sample of workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID, 32 bits
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 0.5);   // keep the CU busy so blocks overlap
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    if (x == 12345.678) out[0] = 0;
}
int main() {
    const int nb = 4096;
    unsigned* d; (void)hipMalloc(&d, nb * 4 * sizeof(unsigned));
    std::vector<unsigned> h(nb * 4);
    probe<<<nb, 256>>>(d, 20000);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<std::string, int> hist;
    for (int b = 0; b < nb; ++b) {
        char buf[64];
        snprintf(buf, sizeof buf, "%u%u%u%u", (h[b * 4] >> 4) & 3, (h[b * 4 + 1] >> 4) & 3, (h[b * 4 + 2] >> 4) & 3, (h[b * 4 + 3] >> 4) & 3);
        hist[buf]++;
    }
    for (auto& kv : hist) printf("simd ids of waves 0..3: %s  x %d\n", kv.first.c_str(), kv.second);
    for (int b = 0; b < 6; ++b) printf("block %d: hw_id %08x %08x %08x %08x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
    return 0;
}
