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

// hipFuncSetAttribute and device-symbol addresses are PER DEVICE, and one process may drive several GPUs
// (hip_state(device) on the Python side): what has been set is remembered per device ordinal.  Relaxed atomics: two
// threads racing on the same device at worst set the same attribute twice.
constexpr int kMaxDevices = 64;
struct PerDeviceSize {
    size_t v[kMaxDevices] = {};
};
// Raises the dynamic-LDS limit of `kernel` on the current device to at least `bytes` (once per device and size).
static inline bool ensure_dynamic_lds(const void *kernel, size_t bytes, PerDeviceSize &state) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    size_t cur = __atomic_load_n(&state.v[dev], __ATOMIC_RELAXED);
    if (bytes <= cur) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    __atomic_store_n(&state.v[dev], bytes, __ATOMIC_RELAXED);
    return true;
}

}  // namespace sgv3d

#define SGV3D_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return sgv3d::fail(SGV3D_EINVAL, __VA_ARGS__); \
    } while (0)
