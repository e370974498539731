// bf16-activation variants of the small layers between the convolutions of HeightNet / MSCThead (bf16 compute mode,
// BASELINE configs[2] / [4]): tensors are bf16 NHWC in HBM, every kernel loads 8 channels (16 bytes) per thread, computes in
// float32 exactly as its f32 twin in misc_layers.hip and rounds once (nearest even) on the store.
//   scale_channels     SELayer gate                               layers/backbones/lss_fpn.py:155-159
//   global_avgpool     ASPP.global_avg_pool                       lss_fpn.py:81-86
//   broadcast_channels F.interpolate of the pooled 1x1 map        lss_fpn.py:101-104
//   upsample_bilinear2x / add_mul_sigmoid   TaskFPN + SABlock     layers/backbones/bsm_lss_fpn.py:151-160, 205-211
//   deform_im2col3x3   mmcv DeformConv2dPack sampling             lss_fpn.py:190-198
// All HBM-bound: algorithmic bytes = the tensors read + written once.
#include "common.hpp"

using namespace sgv3d;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kAvgChunks = 32;

struct F8 {
    float v[8];
};

__device__ __forceinline__ F8 ld8(const __bf16 *p) {
    const bf16x8 q = *reinterpret_cast<const bf16x8 *>(p);
    F8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)q[i];
    return r;
}

__device__ __forceinline__ void st8(__bf16 *p, const F8 &r) {
    const f32x4 lo = {r.v[0], r.v[1], r.v[2], r.v[3]}, hi = {r.v[4], r.v[5], r.v[6], r.v[7]};
    const bf16x4 l4 = __builtin_convertvector(lo, bf16x4), h4 = __builtin_convertvector(hi, bf16x4);
    *reinterpret_cast<bf16x8 *>(p) = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(kBlock) void scale_channels_bf16_kernel(long long total8, int P, int C8, const __bf16 *__restrict__ x,
                                                                     const float *__restrict__ gate, __bf16 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total8) return;
    const int c = (int)(i % C8);
    const long long b = i / ((long long)P * C8);
    F8 v = ld8(x + i * 8);
    const float *g = gate + (b * C8 + c) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] *= g[k];
    st8(y + i * 8, v);
}

// same two deterministic stages as the f32 kernel: partial sums over pixel ranges (f32), then a fixed-order final sum
__global__ __launch_bounds__(kBlock) void global_avgpool_bf16_partial_kernel(int P, int C, int ld, const __bf16 *__restrict__ x,
                                                                             float *__restrict__ part) {
    __shared__ float red[4][64];
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int per = (P + kAvgChunks - 1) / kAvgChunks;
    const int p0 = chunk * per, p1 = min(P, p0 + per);
    float s = 0.f;
    if (c < C)
        for (int p = p0 + g; p < p1; p += 4) s += (float)x[((long long)b * P + p) * ld + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C)
        part[((long long)b * kAvgChunks + chunk) * C + c] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void global_avgpool_bf16_final_kernel(int B, int P, int C, const float *__restrict__ part,
                                                                           float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)B * C) return;
    const int b = (int)(i / C), c = (int)(i - (long long)b * C);
    float s = 0.f;
    for (int k = 0; k < kAvgChunks; ++k) s += part[((long long)b * kAvgChunks + k) * C + c];
    y[i] = s / (float)P;
}

__global__ __launch_bounds__(kBlock) void broadcast_channels_bf16_kernel(long long total8, int P, int C8, int ld, int coff,
                                                                         const float *__restrict__ v, __bf16 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total8) return;
    const int c = (int)(i % C8);
    const long long bp = i / C8;
    const long long b = bp / P;
    F8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = v[(b * C8 + c) * 8 + k];
    st8(y + bp * ld + coff + c * 8, r);
}

__global__ __launch_bounds__(kBlock) void upsample_bilinear2x_bf16_kernel(int B, int H, int W, int C8, const __bf16 *__restrict__ x,
                                                                          __bf16 *__restrict__ y) {
    const long long total = (long long)B * 2 * H * 2 * W * C8;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C8);
    long long t = i / C8;
    const int ow = (int)(t % (2 * W));
    t /= (2 * W);
    const int oh = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const float sh = fmaxf(0.5f * ((float)oh + 0.5f) - 0.5f, 0.f), sw = fmaxf(0.5f * ((float)ow + 0.5f) - 0.5f, 0.f);
    const int h1 = (int)sh, w1 = (int)sw;
    const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
    const float lh1 = sh - (float)h1, lw1 = sw - (float)w1;
    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const __bf16 *p = x + (((long long)b * H * W) * C8 + c) * 8;
    const long long ps = (long long)C8 * 8;
    const F8 a = ld8(p + ((long long)h1 * W + w1) * ps), bq = ld8(p + ((long long)h1 * W + w1 + w1p) * ps);
    const F8 cq = ld8(p + ((long long)(h1 + h1p) * W + w1) * ps), d = ld8(p + ((long long)(h1 + h1p) * W + w1 + w1p) * ps);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = lh0 * (lw0 * a.v[k] + lw1 * bq.v[k]) + lh1 * (lw0 * cq.v[k] + lw1 * d.v[k]);
    st8(y + i * 8, o);
}

__global__ __launch_bounds__(kBlock) void add_mul_sigmoid_bf16_kernel(long long n8, const __bf16 *__restrict__ a,
                                                                      const __bf16 *__restrict__ b, const __bf16 *__restrict__ c,
                                                                      __bf16 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n8) return;
    const F8 av = ld8(a + i * 8), bv = ld8(b + i * 8), cv = ld8(c + i * 8);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = av.v[k] + bv.v[k] * (1.f / (1.f + expf(-cv.v[k])));
    st8(y + i * 8, o);
}

__global__ __launch_bounds__(kBlock) void deform_im2col3x3_bf16_kernel(int B, int H, int W, int C, int groups,
                                                                       const __bf16 *__restrict__ x, const float *__restrict__ off,
                                                                       int off_ld, __bf16 *__restrict__ col) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * 9 * C8;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    long long t = i / C8;
    const int tap = (int)(t % 9);
    t /= 9;  // pixel index b*H*W + h*W + w
    const int w_ = (int)(t % W);
    const long long t2 = t / W;
    const int h_ = (int)(t2 % H);
    const int b = (int)(t2 / H);
    const int ky = tap / 3, kx = tap - ky * 3;
    const float oy = off[t * off_ld + 2 * tap], ox = off[t * off_ld + 2 * tap + 1];
    const float hf = (float)(h_ - 1 + ky) + oy;
    const float wf = (float)(w_ - 1 + kx) + ox;
    F8 val;
#pragma unroll
    for (int k = 0; k < 8; ++k) val.v[k] = 0.f;
    if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
        const int h_low = (int)floorf(hf), w_low = (int)floorf(wf);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = hf - (float)h_low, lw = wf - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const __bf16 *base = x + (long long)b * H * W * C + c8 * 8;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        // same expression order as the f32 kernel: w1 v1 + w2 v2 + w3 v3 + w4 v4 with absent corners contributing 0
        F8 v1, v2, v3, v4;
#pragma unroll
        for (int k = 0; k < 8; ++k) v1.v[k] = v2.v[k] = v3.v[k] = v4.v[k] = 0.f;
        if (h_low >= 0 && w_low >= 0) v1 = ld8(base + ((long long)h_low * W + w_low) * C);
        if (h_low >= 0 && w_high <= W - 1) v2 = ld8(base + ((long long)h_low * W + w_high) * C);
        if (h_high <= H - 1 && w_low >= 0) v3 = ld8(base + ((long long)h_high * W + w_low) * C);
        if (h_high <= H - 1 && w_high <= W - 1) v4 = ld8(base + ((long long)h_high * W + w_high) * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) val.v[k] = w1 * v1.v[k] + w2 * v2.v[k] + w3 * v3.v[k] + w4 * v4.v[k];
    }
    const int cpg = C / groups;
    const int c = c8 * 8;
    const int g = c / cpg, cg = c - g * cpg;
    st8(col + ((t * groups + g) * 9 + tap) * cpg + cg, val);
}

}  // namespace

extern "C" int sgv3d_scale_channels_bf16(int batch, int pixels, int channels, const void *x, const float *gate, void *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && (channels & 7) == 0, "scale_channels_bf16: channels must be a multiple of 8");
    SGV3D_REQUIRE(x && gate && y, "scale_channels_bf16: null pointer");
    const long long total8 = (long long)batch * pixels * (channels / 8);
    hipLaunchKernelGGL(scale_channels_bf16_kernel, dim3(cdiv(total8, kBlock)), dim3(kBlock), 0, as_stream(stream), total8, pixels,
                       channels / 8, static_cast<const __bf16 *>(x), gate, static_cast<__bf16 *>(y));
    return check_launch("scale_channels_bf16_kernel");
}

extern "C" int sgv3d_global_avgpool_bf16(int batch, int pixels, int channels, int x_ld, const void *x, float *y, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && x_ld >= channels, "global_avgpool_bf16: bad shape");
    SGV3D_REQUIRE(x && y && workspace, "global_avgpool_bf16: null pointer");
    if (workspace_bytes < sizeof(float) * (size_t)batch * kAvgChunks * channels)
        return fail(SGV3D_ENOSPACE, "global_avgpool_bf16: workspace too small (sgv3d_global_avgpool_workspace_bytes)");
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(global_avgpool_bf16_partial_kernel, dim3(cdiv(channels, 64), kAvgChunks, batch), dim3(kBlock), 0,
                       as_stream(stream), pixels, channels, x_ld, static_cast<const __bf16 *>(x), part);
    hipLaunchKernelGGL(global_avgpool_bf16_final_kernel, dim3(cdiv((long long)batch * channels, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), batch, pixels, channels, part, y);
    return check_launch("global_avgpool_bf16_kernel");
}

extern "C" int sgv3d_broadcast_channels_bf16(int batch, int pixels, int channels, int y_ld, int y_coff, const float *v, void *y,
                                             void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && y_ld >= y_coff + channels && y_coff >= 0, "broadcast_channels_bf16: bad shape");
    SGV3D_REQUIRE(((channels | y_ld | y_coff) & 7) == 0, "broadcast_channels_bf16: channels / y_ld / y_coff must be multiples of 8");
    SGV3D_REQUIRE(v && y, "broadcast_channels_bf16: null pointer");
    const long long total8 = (long long)batch * pixels * (channels / 8);
    hipLaunchKernelGGL(broadcast_channels_bf16_kernel, dim3(cdiv(total8, kBlock)), dim3(kBlock), 0, as_stream(stream), total8, pixels,
                       channels / 8, y_ld, y_coff, v, static_cast<__bf16 *>(y));
    return check_launch("broadcast_channels_bf16_kernel");
}

extern "C" int sgv3d_upsample_bilinear2x_bf16(int batch, int h, int w, int channels, const void *x, void *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0 && (channels & 7) == 0, "upsample_bilinear2x_bf16: channels must be a multiple of 8");
    SGV3D_REQUIRE(x && y, "upsample_bilinear2x_bf16: null pointer");
    const long long total = (long long)batch * 4 * h * w * (channels / 8);
    hipLaunchKernelGGL(upsample_bilinear2x_bf16_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, h, w,
                       channels / 8, static_cast<const __bf16 *>(x), static_cast<__bf16 *>(y));
    return check_launch("upsample_bilinear2x_bf16_kernel");
}

extern "C" int sgv3d_add_mul_sigmoid_bf16(long long n, const void *a, const void *b, const void *c, void *y, void *stream) {
    SGV3D_REQUIRE(n > 0 && (n & 7) == 0, "add_mul_sigmoid_bf16: n must be a positive multiple of 8");
    SGV3D_REQUIRE(a && b && c && y, "add_mul_sigmoid_bf16: null pointer");
    hipLaunchKernelGGL(add_mul_sigmoid_bf16_kernel, dim3(cdiv(n / 8, kBlock)), dim3(kBlock), 0, as_stream(stream), n / 8,
                       static_cast<const __bf16 *>(a), static_cast<const __bf16 *>(b), static_cast<const __bf16 *>(c),
                       static_cast<__bf16 *>(y));
    return check_launch("add_mul_sigmoid_bf16_kernel");
}

extern "C" int sgv3d_deform_im2col3x3_bf16(int batch, int h, int w, int channels, int groups, const void *x, const float *offset,
                                           int off_ld, void *col, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0 && groups > 0 && channels % groups == 0 &&
                      ((channels / groups) & 7) == 0 && off_ld >= 18,
                  "deform_im2col3x3_bf16: bad shape (channels per group must be a multiple of 8)");
    SGV3D_REQUIRE(x && offset && col, "deform_im2col3x3_bf16: null pointer");
    const long long total = (long long)batch * h * w * 9 * (channels / 8);
    hipLaunchKernelGGL(deform_im2col3x3_bf16_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, h, w,
                       channels, groups, static_cast<const __bf16 *>(x), offset, off_ld, static_cast<__bf16 *>(col));
    return check_launch("deform_im2col3x3_bf16_kernel");
}
