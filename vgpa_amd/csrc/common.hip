// Error bookkeeping and version query of libvgpa_hip.so.
#include "common.h"

#include <cstdarg>
#include <cstdio>

namespace vgpa {

static thread_local char g_last_error[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VGPA_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return VGPA_OK;
}

}  // namespace vgpa

extern "C" int vgpa_version(void) { return VGPA_ABI_VERSION; }

extern "C" const char* vgpa_last_error_string(void) { return vgpa::g_last_error; }
