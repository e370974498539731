#include "common.hpp"

#include <cstdarg>
#include <cstdio>

namespace sgv3d {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SGV3D_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SGV3D_OK;
}

const char *last_error() { return g_err; }

}  // namespace sgv3d

extern "C" const char *sgv3d_last_error(void) { return sgv3d::last_error(); }
extern "C" int sgv3d_abi_version(void) { return 1; }
