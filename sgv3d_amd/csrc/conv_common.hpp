// Shared by the convolution kernels (conv_igemm.hip, conv_wino.hip): launch arguments, the common
// epilogue (folded BN / bias, residual, ReLU, SE gate, store in the mode's layout) and the split-K tail.
#pragma once
#include "common.hpp"

namespace sgv3d {

struct ConvArgs {
    const float *x, *w, *scale, *bias, *res, *gate, *zeros;
    float *y;
    int M, N, K, k_pad;
    int in_h, in_w, cin, out_h, out_w, cout;
    int m_h, m_w;  // spatial dims used to decode m (output dims; input dims for the deconv GEMM)
    int kh, kw, stride, pad, dil;
    int x_ld, x_coff, y_ld, y_coff, res_ld, relu, mode, ks;   // mode: SGV3D_CONV_* | kConvYBf16 | kConvResBf16
    int tiles_m, tiles_n;
    int wb_y, wb_x;  // Winograd kernel: 16x16-pixel output blocks per image.  f32 implicit GEMM: wb_y > 0 = grouped GEMM
                     // (conv_gemm_grouped): rows [g * wb_y, (g + 1) * wb_y) multiply weight block g (wb_x floats apart)
    unsigned x_bytes, w_bytes;   // implicit GEMM: sizes of the input / packed-weight buffers (buffer resources)
    int korder;  // 0: k = tap*cin + ci   1: k = (ci/32 * taps + tap)*32 + ci%32  (cin % 32 == 0)
    int split_k; // > 1: blockIdx.y owns a slice of the k-tiles and stores raw partial sums to ws
    float *ws;   // [split_k][M][N] partial sums (split_k > 1)
};

// bf16-activation mode (sgv3d_conv2d_forward_bf16io): y / res point to bf16 tensors.  Carried in the high bits of `mode`:
// two more kernel-argument words pushed the 128x128 bf16 kernel (256 VGPRs at two workgroups per CU) into spilling.
constexpr int kConvModeMask = 0xff, kConvYBf16 = 0x100, kConvResBf16 = 0x200;

// Shared by the conv kernel (split_k == 1) and the split-K reduce kernel: scale/shift (folded BN or
// bias), residual, ReLU, SE gate and the store in the mode's layout.
static __device__ __forceinline__ void conv_epilogue_store(const ConvArgs &a, int row, int col, float accv) {
    const int hw = a.m_h * a.m_w;
    const int mode = a.mode & kConvModeMask;
    int co = col, dy = 0, dx = 0;
    if (mode == SGV3D_CONV_DECONV) {
        const int tap = col / a.cout;
        co = col - tap * a.cout;
        dy = tap / a.ks;
        dx = tap - dy * a.ks;
    }
    float v = accv * (a.scale ? a.scale[co] : 1.f) + (a.bias ? a.bias[co] : 0.f);
    size_t yi;
    int img = 0;
    if (mode == SGV3D_CONV_NORMAL) {
        yi = (size_t)row * a.y_ld + a.y_coff + co;
        if (a.gate) img = row / hw;
    } else {
        img = row / hw;
        const int pix = row - img * hw;
        if (mode == SGV3D_CONV_DECONV) {
            const int ih = pix / a.m_w, iw = pix - ih * a.m_w;
            yi = ((size_t)(img * a.out_h + ih * a.ks + dy) * a.out_w + (iw * a.ks + dx)) * a.y_ld + a.y_coff + co;
        } else if (mode == SGV3D_CONV_NCHW_OUT) {
            yi = ((size_t)img * a.y_ld + a.y_coff + co) * hw + pix;
        } else {  // GROUP_PLANES: [cout/g][M][g], g = a.ks
            const int grp = co / a.ks;
            yi = ((size_t)grp * a.M + row) * a.ks + (co - grp * a.ks);
        }
    }
    if (a.res)
        v += (a.mode & kConvResBf16) ? (float)reinterpret_cast<const __bf16 *>(a.res)[(size_t)row * a.res_ld + co] : a.res[(size_t)row * a.res_ld + co];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.gate) v *= a.gate[(size_t)img * a.cout + co];
    if (a.mode & kConvYBf16) reinterpret_cast<__bf16 *>(a.y)[yi] = (__bf16)v;
    else a.y[yi] = v;
}

// Split-K second stage (conv_igemm.hip): sums a.ws[split][M][N] in fixed order and runs the epilogue.
int launch_splitk_reduce(const ConvArgs &a, hipStream_t st);
// 16 zero bytes in device memory that padded / out-of-image lanes load from (conv_igemm.hip).
const float *conv_zero_block();
// Grouped f32 GEMM on the implicit-GEMM kernel (conv_igemm.hip), used by the F(4x4) Winograd path (conv_wino4.hip):
// y[g * rows + r][n] = sum_k x[g * rows + r][k] * w_g[n][k] for g < groups, rows % 64 == 0, K % 4 == 0; w: `groups` packed
// weight blocks [cout_pad][k_pad] (sgv3d_conv_pack_weight of a 1x1 convolution), k_order as packed.  tile: SGV3D_TILE_64x64
// or SGV3D_TILE_64x128 (the m-tiles must not straddle groups).
int conv_gemm_grouped(const float *x, const float *w, float *y, int rows, int groups, int K, int N, int k_pad, int cout_pad,
                      int k_order, int tile, hipStream_t st);
// the same product on v_mfma_f32_16x16x4_f32 with 48 x 64 workgroup tiles (gemm16_grouped.hip): rows % 48 == 0, K % 32 == 0
int conv_gemm_grouped16(const float *x, const float *w, float *y, int rows, int groups, int K, int N, int k_pad, int cout_pad,
                        hipStream_t st);
// f32x3 position GEMM of the F(4x4) path on bf16 planes (gemm_x3_grouped.hip); variant = m-tile index {48, 64, 96, 112, 128 rows}
// + 5 for the five-wave (160-column) form
int conv_gemm_grouped_x3(const void *x, const void *w, float *y, int rows, int K, int N, int cout_pad, int variant, hipStream_t st,
                         size_t w_block_stride);
int gemm_x3_tile_rows(int variant);

}  // namespace sgv3d
