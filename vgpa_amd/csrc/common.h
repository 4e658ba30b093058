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

}  // namespace vgpa

#define VGPA_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::vgpa::fail(VGPA_EINVAL, __VA_ARGS__); \
    } while (0)
