// Lift: softmax over the D height bins, outer product with the C context channels, written
// channel-last.  Reference: layers/backbones/lss_fpn.py:462-466 (softmax, unsqueeze/multiply) and the
// permute(0,1,3,4,5,2)+contiguous of :486,:490 — here one streaming pass, HBM-write bound
// (4*D*P*C bytes out, 4*P*(D+C) in).
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kPixTile = 16;   // pixels per workgroup
constexpr int kBlock = 256;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// height_context [B, P, D+C]; prob [B, D, P]; lifted [B, D, P, C] (f32, or bf16 with LB: bf16 compute mode, C % 4 == 0 --
// the products are formed in f32 and rounded once; the tensor is the largest HBM stream of the path, 2x smaller this way)
template <bool LB>
__global__ __launch_bounds__(kBlock) void lift_kernel(int P, int D, int C, const float *__restrict__ hc,
                                                      float *__restrict__ prob, void *__restrict__ lifted_) {
    float *const lifted = static_cast<float *>(lifted_);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *pr = smem;                      // [kPixTile][D]
    float *cx = smem + kPixTile * D;       // [kPixTile][C]   (D*kPixTile is a multiple of 4 floats)
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * kPixTile;
    const int npix = min(kPixTile, P - p0);
    const int ld = D + C;
    const float *src = hc + ((size_t)b * P + p0) * ld;
    for (int i = threadIdx.x; i < npix * ld; i += kBlock) {
        const int px = i / ld, ch = i - px * ld;
        const float v = src[i];
        if (ch < D) pr[px * D + ch] = v;
        else cx[px * C + (ch - D)] = v;
    }
    __syncthreads();
    // softmax per pixel: one 16-lane group per pixel (16 pixels x 16 lanes = 256 threads)
    {
        const int px = threadIdx.x >> 4, l = threadIdx.x & 15;
        float m = -INFINITY;
        if (px < npix)
            for (int d = l; d < D; d += 16) m = fmaxf(m, pr[px * D + d]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
        float s = 0.f;
        if (px < npix)
            for (int d = l; d < D; d += 16) {
                const float e = expf(pr[px * D + d] - m);
                pr[px * D + d] = e;
                s += e;
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (px < npix)
            for (int d = l; d < D; d += 16) pr[px * D + d] = pr[px * D + d] / s;
    }
    __syncthreads();
    if (prob) {
        for (int i = threadIdx.x; i < D * npix; i += kBlock) {
            const int d = i / npix, px = i - d * npix;
            prob[((size_t)b * D + d) * P + p0 + px] = pr[px * D + d];
        }
    }
    if (lifted) {
        if ((C & 3) == 0) {
            const int c4n = C >> 2;
            const int per_d = npix * c4n;
            for (int i = threadIdx.x; i < D * per_d; i += kBlock) {
                const int d = i / per_d, r = i - d * per_d;
                const int px = r / c4n, c4 = r - px * c4n;
                const float w = pr[px * D + d];
                const float4 v = *reinterpret_cast<const float4 *>(cx + px * C + c4 * 4);
                const size_t at = (((size_t)b * D + d) * P + p0 + px) * C + c4 * 4;
                if constexpr (LB) {
                    const f32x4 o = {w * v.x, w * v.y, w * v.z, w * v.w};
                    *reinterpret_cast<bf16x4 *>(static_cast<__bf16 *>(lifted_) + at) = __builtin_convertvector(o, bf16x4);
                } else {
                    *reinterpret_cast<float4 *>(lifted + at) = make_float4(w * v.x, w * v.y, w * v.z, w * v.w);
                }
            }
        } else {
            const int per_d = npix * C;
            for (int i = threadIdx.x; i < D * per_d; i += kBlock) {
                const int d = i / per_d, r = i - d * per_d;
                const int px = r / C, c = r - px * C;
                lifted[(((size_t)b * D + d) * P + p0 + px) * C + c] = pr[px * D + d] * cx[px * C + c];
            }
        }
    }
}

}  // namespace

extern "C" int sgv3d_lift(int batch_size, int num_pixels, int num_depth, int num_channels,
                          const float *height_context, float *prob, float *lifted, void *stream) {
    SGV3D_REQUIRE(batch_size > 0 && num_pixels > 0 && num_depth > 0 && num_channels > 0, "lift: non-positive size");
    SGV3D_REQUIRE(height_context && (prob || lifted), "lift: null pointer");
    // float4 path needs the context block in LDS 16-B aligned: kPixTile*D floats is a multiple of 4.
    const size_t lds = sizeof(float) * (size_t)kPixTile * (num_depth + num_channels);
    SGV3D_REQUIRE(lds <= 64 * 1024, "lift: D+C=%d too large for the LDS tile", num_depth + num_channels);
    if ((num_channels & 3) == 0)
        SGV3D_REQUIRE(lifted == nullptr || (reinterpret_cast<uintptr_t>(lifted) & 15) == 0, "lift: lifted must be 16-B aligned");
    dim3 grid(cdiv(num_pixels, kPixTile), batch_size);
    hipLaunchKernelGGL(lift_kernel<false>, grid, dim3(kBlock), lds, as_stream(stream), num_pixels, num_depth, num_channels,
                       height_context, prob, static_cast<void *>(lifted));
    return check_launch("lift_kernel");
}

extern "C" int sgv3d_lift_bf16(int batch_size, int num_pixels, int num_depth, int num_channels, const float *height_context,
                               float *prob, void *lifted_bf16, void *stream) {
    SGV3D_REQUIRE(batch_size > 0 && num_pixels > 0 && num_depth > 0 && num_channels > 0 && (num_channels & 3) == 0,
                  "lift_bf16: sizes must be positive and the channel count a multiple of 4");
    SGV3D_REQUIRE(height_context && lifted_bf16, "lift_bf16: null pointer");
    const size_t lds = sizeof(float) * (size_t)kPixTile * (num_depth + num_channels);
    SGV3D_REQUIRE(lds <= 64 * 1024, "lift_bf16: D+C=%d too large for the LDS tile", num_depth + num_channels);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(lifted_bf16) & 7) == 0, "lift_bf16: lifted must be 8-B aligned");
    dim3 grid(cdiv(num_pixels, kPixTile), batch_size);
    hipLaunchKernelGGL(lift_kernel<true>, grid, dim3(kBlock), lds, as_stream(stream), num_pixels, num_depth, num_channels,
                       height_context, prob, lifted_bf16);
    return check_launch("lift_kernel<bf16>");
}
