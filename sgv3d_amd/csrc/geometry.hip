// Frustum -> voxel-index geometry for gfx950.
// Reference: LSSFPN.get_geometry / height2localtion and the quantise expression,
//   layers/backbones/lss_fpn.py:350-401, :487-488.
//
// Parity contract: float32 with ONE rounding per operation in exactly the order of
// oracle/geometry_ref.py (which is pinned bit-for-bit against the reference's torch-CPU run), so this
// translation unit is compiled with -ffp-contract=off and the per-point sums are written as explicit
// mul/add chains.  Division is hipcc's default correctly-rounded fp32 divide.
#include "common.hpp"

#pragma clang fp contract(off)

using namespace sgv3d;

namespace {

// ---- 4x4 helpers (mirror oracle/geometry_ref.py::inv4 / mm4 operation for operation) -------------
__device__ void inv4(const float *__restrict__ src, float *__restrict__ dst) {
    float A[4][4];
    int piv[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) A[i][j] = src[i * 4 + j];
    for (int j = 0; j < 4; ++j) {
        int p = j;
        float best = fabsf(A[j][j]);
        for (int i = j + 1; i < 4; ++i) {
            const float a = fabsf(A[i][j]);
            if (a > best) { best = a; p = i; }
        }
        if (p != j) {
            for (int c = 0; c < 4; ++c) { const float t = A[j][c]; A[j][c] = A[p][c]; A[p][c] = t; }
            const int t = piv[j]; piv[j] = piv[p]; piv[p] = t;
        }
        const float r = 1.0f / A[j][j];
        for (int i = j + 1; i < 4; ++i) A[i][j] = A[i][j] * r;
        for (int jj = j + 1; jj < 4; ++jj)
            for (int i = j + 1; i < 4; ++i) {
                const float prod = A[i][j] * A[j][jj];
                A[i][jj] = A[i][jj] - prod;
            }
    }
    float X[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) X[i][j] = (piv[i] == j) ? 1.0f : 0.0f;
    float dinv[4];
    for (int k = 0; k < 4; ++k) dinv[k] = 1.0f / A[k][k];
    for (int c = 0; c < 4; ++c) {
        for (int k = 0; k < 4; ++k)
            for (int i = k + 1; i < 4; ++i) {
                const float prod = X[k][c] * A[i][k];
                X[i][c] = X[i][c] - prod;
            }
        for (int k = 3; k >= 0; --k) {
            X[k][c] = X[k][c] * dinv[k];
            for (int i = 0; i < k; ++i) {
                const float prod = X[k][c] * A[i][k];
                X[i][c] = X[i][c] - prod;
            }
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) dst[i * 4 + j] = X[i][j];
}

__device__ void mm4(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) {
                const float prod = A[i * 4 + k] * B[k * 4 + j];
                acc = acc + prod;
            }
            C[i * 4 + j] = acc;
        }
}

// run: nullable DEVICE flag; 0 = the calibration is the one `prep` / the index tensor were computed from (sgv3d_calib_changed):
// the launch returns at once and leaves its outputs as they are
__global__ void calib_prep_kernel(int num_cams, const float *__restrict__ s2e, const float *__restrict__ s2v,
                                  const float *__restrict__ intrin, const float *__restrict__ ida,
                                  float *__restrict__ prep, const int *__restrict__ run) {
    if (run && *run == 0) return;
    const int cam = blockIdx.x * blockDim.x + threadIdx.x;
    if (cam >= num_cams) return;
    float tmp[16];
    float *o = prep + (size_t)cam * 48;
    inv4(ida + cam * 16, o);                 // lss_fpn.py:390
    inv4(intrin + cam * 16, tmp);
    mm4(s2v + cam * 16, tmp, o + 16);        // lss_fpn.py:361
    inv4(s2v + cam * 16, tmp);
    mm4(s2e + cam * 16, tmp, o + 32);        // lss_fpn.py:367
}

// y = M @ v, sequential sum from 0 (torch's small-bmm order)
__device__ __forceinline__ void mv4(const float *__restrict__ M, const float v[4], float out[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float prod = M[i * 4 + k] * v[k];
            acc = acc + prod;
        }
        out[i] = acc;
    }
}

// float -> int32 with the semantics the reference's `.int()` has on a GPU tensor:
// truncate toward zero, saturate, NaN -> 0.
__device__ __forceinline__ int cvt_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}

struct GeomConst {
    float origin[3];  // voxel_coord - voxel_size / 2
    float vsize[3];
};

__global__ __launch_bounds__(256) void geometry_kernel(int pts_per_cam, int cams_per_batch,
                                                       const float4 *__restrict__ frustum,
                                                       const float *__restrict__ prep,
                                                       const float *__restrict__ ref_h,
                                                       const float *__restrict__ bda, GeomConst gc,
                                                       int32_t *__restrict__ geom_xyz,
                                                       float *__restrict__ geom_f, const int *__restrict__ run) {
    if (run && *run == 0) return;
    __shared__ float M[64];  // ida_inv | combine_virtual | combine_ego | bda
    const int cam = blockIdx.y;
    if (threadIdx.x < 48) M[threadIdx.x] = prep[(size_t)cam * 48 + threadIdx.x];
    if (bda && threadIdx.x >= 48 && threadIdx.x < 64)
        M[threadIdx.x] = bda[(size_t)(cam / cams_per_batch) * 16 + (threadIdx.x - 48)];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pts_per_cam) return;
    const float4 f = frustum[p];
    const float v[4] = {f.x, f.y, f.z, f.w};
    float q[4], pc[4], pv[4], e[4];
    mv4(M, v, q);                                        // :390  ida^-1 @ (u, v, d, 1)
    const float neg = -1.0f * q[2];
    const float height = neg + ref_h[cam];               // :354
    pc[0] = q[0] * 10.0f;                                // :356-360
    pc[1] = q[1] * 10.0f;
    pc[2] = 10.0f;
    pc[3] = q[3];
    mv4(M + 16, pc, pv);                                 // :362
    const float ratio = height / pv[1];                  // :363
    float r[4];
    r[0] = pv[0] * ratio;                                // :365
    r[1] = pv[1] * ratio;
    r[2] = pv[2] * ratio;
    r[3] = 1.0f;                                         // :366
    mv4(M + 32, r, e);                                   // :368-369
    if (bda) {                                           // :394-398
        float t[4];
        mv4(M + 48, e, t);
        e[0] = t[0]; e[1] = t[1]; e[2] = t[2];
    }
    const size_t o = ((size_t)cam * pts_per_cam + p) * 3;
    if (geom_f) {
        geom_f[o + 0] = e[0];
        geom_f[o + 1] = e[1];
        geom_f[o + 2] = e[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {                        // :487-488
        const float d = e[a] - gc.origin[a];
        const float qn = d / gc.vsize[a];
        geom_xyz[o + a] = cvt_i32(qn);
    }
}

}  // namespace

extern "C" int sgv3d_calib_prep_gated(int num_cams, const float *sensor2ego, const float *sensor2virtual,
                                      const float *intrin, const float *ida, float *prep, const int32_t *run, void *stream) {
    SGV3D_REQUIRE(num_cams > 0, "calib_prep: num_cams=%d", num_cams);
    SGV3D_REQUIRE(sensor2ego && sensor2virtual && intrin && ida && prep, "calib_prep: null pointer");
    hipLaunchKernelGGL(calib_prep_kernel, dim3(cdiv(num_cams, 64)), dim3(64), 0, as_stream(stream), num_cams,
                       sensor2ego, sensor2virtual, intrin, ida, prep, run);
    return check_launch("calib_prep_kernel");
}

extern "C" int sgv3d_calib_prep(int num_cams, const float *sensor2ego, const float *sensor2virtual,
                                const float *intrin, const float *ida, float *prep, void *stream) {
    return sgv3d_calib_prep_gated(num_cams, sensor2ego, sensor2virtual, intrin, ida, prep, nullptr, stream);
}

// ---- "is this the calibration the cached state was computed from?" decided on the device --------------------------------
namespace {
struct CalibSrc {
    const unsigned *p[8];
    int words[8];
    int n;
};
// One workgroup: compares the calibration tensors word for word with the copy taken at the last change; on a difference (or
// force) stores the new values as the copy.  changed[0] = 1 / 0.
__global__ __launch_bounds__(256) void calib_changed_kernel(CalibSrc s, unsigned *__restrict__ copy, int force, int *__restrict__ changed) {
    __shared__ int diff;
    if (threadIdx.x == 0) diff = force ? 1 : 0;
    __syncthreads();
    int off = 0, mine = 0;
    for (int t = 0; t < s.n; ++t) {
        for (int i = threadIdx.x; i < s.words[t]; i += 256) mine |= s.p[t][i] != copy[off + i];
        off += s.words[t];
    }
    if (mine) atomicOr(&diff, 1);
    __syncthreads();
    const int d = diff;
    if (d) {
        off = 0;
        for (int t = 0; t < s.n; ++t) {
            for (int i = threadIdx.x; i < s.words[t]; i += 256) copy[off + i] = s.p[t][i];
            off += s.words[t];
        }
    }
    if (threadIdx.x == 0) changed[0] = d;
}
}  // namespace

extern "C" int sgv3d_calib_changed(int n, const void *const *tensors, const int32_t *nbytes, void *copy, int force,
                                   int32_t *changed, void *stream) {
    SGV3D_REQUIRE(n > 0 && n <= 8 && tensors && nbytes && copy && changed, "calib_changed: 1..8 tensors, non-null pointers");
    CalibSrc s;
    s.n = n;
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(tensors[i] && nbytes[i] > 0 && nbytes[i] % 4 == 0 && (reinterpret_cast<uintptr_t>(tensors[i]) & 3) == 0,
                      "calib_changed: tensor %d must be non-null, 4-byte aligned, a multiple of 4 bytes", i);
        s.p[i] = static_cast<const unsigned *>(tensors[i]);
        s.words[i] = nbytes[i] / 4;
    }
    hipLaunchKernelGGL(calib_changed_kernel, dim3(1), dim3(256), 0, as_stream(stream), s, static_cast<unsigned *>(copy), force, changed);
    return check_launch("calib_changed_kernel");
}

extern "C" int sgv3d_geometry_voxel_index_gated(int num_cams, int cams_per_batch, int num_depth, int feat_h,
                                                int feat_w, const float *frustum, const float *prep,
                                                const float *ref_h, const float *bda, const float *voxel_coord,
                                                const float *voxel_size, int32_t *geom_xyz, float *geom_f,
                                                const int32_t *run, void *stream);

extern "C" int sgv3d_geometry_voxel_index(int num_cams, int cams_per_batch, int num_depth, int feat_h,
                                          int feat_w, const float *frustum, const float *prep,
                                          const float *ref_h, const float *bda, const float *voxel_coord,
                                          const float *voxel_size, int32_t *geom_xyz, float *geom_f,
                                          void *stream) {
    return sgv3d_geometry_voxel_index_gated(num_cams, cams_per_batch, num_depth, feat_h, feat_w, frustum, prep, ref_h, bda,
                                            voxel_coord, voxel_size, geom_xyz, geom_f, nullptr, stream);
}

extern "C" int sgv3d_geometry_voxel_index_gated(int num_cams, int cams_per_batch, int num_depth, int feat_h,
                                                int feat_w, const float *frustum, const float *prep,
                                                const float *ref_h, const float *bda, const float *voxel_coord,
                                                const float *voxel_size, int32_t *geom_xyz, float *geom_f,
                                                const int32_t *run, void *stream) {
    SGV3D_REQUIRE(num_cams > 0 && cams_per_batch > 0 && num_depth > 0 && feat_h > 0 && feat_w > 0,
                  "geometry_voxel_index: non-positive size");
    SGV3D_REQUIRE(frustum && prep && ref_h && voxel_coord && voxel_size && geom_xyz,
                  "geometry_voxel_index: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(frustum) & 15) == 0, "geometry_voxel_index: frustum must be 16-B aligned");
    const long long pts = (long long)num_depth * feat_h * feat_w;
    SGV3D_REQUIRE(pts * num_cams < 0x7fffffffLL / 3, "geometry_voxel_index: too many points");
    GeomConst gc;
    for (int a = 0; a < 3; ++a) {
        // voxel_coord - voxel_size / 2.0 evaluated in float32 like the torch buffers (lss_fpn.py:487)
        volatile float half = voxel_size[a] / 2.0f;
        volatile float org = voxel_coord[a] - half;
        gc.origin[a] = org;
        gc.vsize[a] = voxel_size[a];
    }
    dim3 grid(cdiv(pts, 256), num_cams);
    hipLaunchKernelGGL(geometry_kernel, grid, dim3(256), 0, as_stream(stream), (int)pts, cams_per_batch,
                       reinterpret_cast<const float4 *>(frustum), prep, ref_h, bda, gc, geom_xyz, geom_f, run);
    return check_launch("geometry_kernel");
}
