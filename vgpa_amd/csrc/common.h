// Shared host-side helpers for the libvgpa_hip.so translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "vgpa.h"

namespace vgpa {

// Record a formatted message as the calling thread's last error and return `code`.
int fail(int code, const char* fmt, ...);

// Check hipGetLastError() after a launch; returns VGPA_OK or VGPA_ELAUNCH.
int check_launch(const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Accesses of data that is touched once per kernel (the (P, N, D, D) arrays stream through: nothing
// of them is in a cache when the next kernel wants it).  VGPA_NT selects the cache policy:
// 0 plain, 1 non-temporal stores, 2 non-temporal loads and stores.  Used by the D = 40 scan kernels
// (measured at 512 problems: separate backward scan 7.6 -> 6.9 ms, forward scan 5.6 -> 5.5 ms, the fused
// backward scan + gradient unchanged); the Lorenz-96 energy kernel keeps plain accesses (it lost 1.5 %).
// Only for accesses that cover whole cache lines per wavefront instruction: the 8-byte-per-lane stores of dL/dA written
// non-temporally raised the fused kernel's HBM write traffic by 8 % (partial lines are not merged), so they stay plain.
#ifndef VGPA_NT
#define VGPA_NT 2
#endif
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_once(const double2* p) {
#if VGPA_NT >= 2
    const f64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f64x2_t*>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ double ld_once(const double* p) {
#if VGPA_NT >= 2
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_once(double2* p, double2 v) {
#if VGPA_NT >= 1
    f64x2_t w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<f64x2_t*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_once(double* p, double v) {
#if VGPA_NT >= 1
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

}  // namespace vgpa

#define VGPA_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::vgpa::fail(VGPA_EINVAL, __VA_ARGS__); \
    } while (0)
