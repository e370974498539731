// Shared host-side helpers for libsgv3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgv3d_hip.h"

namespace sgv3d {

// Records a message for sgv3d_last_error() and returns `code`.
int fail(int code, const char *fmt, ...);
// hipGetLastError() after a launch -> SGV3D_OK / SGV3D_ELAUNCH.
int check_launch(const char *what);

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace sgv3d

#define SGV3D_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return sgv3d::fail(SGV3D_EINVAL, __VA_ARGS__); \
    } while (0)
