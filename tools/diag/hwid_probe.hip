// Which SIMD and compute unit each wavefront of a workgroup lands on (HW_ID, XCC_ID hardware registers).
//   hipcc --offload-arch=gfx950 -O2 -o tools/diag/hwid_probe tools/diag/hwid_probe.hip && tools/diag/hwid_probe [threads] [lds_bytes] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * nw + wave) * 2] = hw;
        out[(blockIdx.x * nw + wave) * 2 + 1] = xcc;
    }
    // stay resident for a while so that the whole grid is in flight together
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if (x == 12345.0) lds[threadIdx.x] = x;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 512;
    const int lds = argc > 2 ? atoi(argv[2]) : 67000;
    const int blocks = argc > 3 ? atoi(argv[3]) : 512;
    const int nw = threads / 64;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * blocks * nw);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 200000);
    std::vector<unsigned> h(2 * blocks * nw);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[(b * nw) * 2], xcc = h[(b * nw) * 2 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        if (b < 6 || b == 256 || b == 257) {
            printf("block %3d  xcc %u se %u sh %u cu %2u   simd of waves:", b, xcc, se, sh, cu);
            for (int w = 0; w < nw; ++w) printf(" %u", (h[(b * nw + w) * 2] >> 4) & 3);
            printf("   wave slots:");
            for (int w = 0; w < nw; ++w) printf(" %u", h[(b * nw + w) * 2] & 0xf);
            printf("\n");
        }
    }
    printf("%zu distinct compute units; blocks sharing a compute unit (first 12):\n", by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 12) break;
        printf("  xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    return 0;
}
