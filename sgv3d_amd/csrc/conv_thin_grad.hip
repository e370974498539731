// Backward of n "thin" 3x3 convolutions in ONE launch per gradient kind (gfx950): stride 1, dilation 1, cin -> 1..4 output
// channels -- the 36 final layers of the CenterHead branches (64 -> 1 / 2 / 3 at 256 x 256; reference:
// layers/heads/bev_height_head.py:75-110 through mmdet3d SeparateHead, whose backward the reference leaves to cuDNN, one call
// per layer and gradient).  Per layer the work is one pass over a 33 MB hidden map with a handful of FMAs per element: alone a
// launch is a few microseconds of memory time behind ~15 us of latency (two units per wave, nothing to overlap with), and the
// MFMA kernels run these shapes at 1-6 TFLOP/s (an MFMA tile would be 94-98 % padding).  Batched over the layers
// (blockIdx.y / blockIdx.z = layer) every SIMD holds 4 waves of independent work:
//
//   thin_wgrad_kernel   dW[c][ci][r][s] = sum_p dY[p][c] X[p + (r, s) - pad][ci]  and  db[c] = sum_p dY[p][c]
//                       a lane owns one input channel, a wave walks units of 3 output rows x 32 pixels: the 5 x 18 input values of its
//                       channel in registers (coalesced 256-byte rows), the dY values of 8 pixels of a row in ONE coalesced load, broadcast
//                       with v_readlane, 9 x COUT (+ COUT for the bias) FMAs per pixel.  Waves meet in LDS in wave order, one
//                       partial set per workgroup in the workspace, thin_reduce_kernel adds them in workgroup order
//                       (deterministic) into OIHW / the bias gradient.
//   thin_dgrad_kernel   dX[p][ci] = sum_{r,s,c} dY[p + pad - (r, s)][c] W[c][ci][r][s]
//                       a lane owns 4 consecutive input channels of 4 consecutive pixels (16 lanes = one 256-byte pixel row of 64
//                       channels) of 2 rows, its 9 x COUT x 4 weights in registers for the whole launch, the 4 x 6 x COUT dY values of a
//                       unit from L1 / L2 (dY of a layer is 0.5-1.5 MB), 16-byte stores.  No LDS, no barrier.
//
//   thin_fwd_kernel     the forward of the same layers (cin <= 64), f32 FMAs + a 16-lane DPP reduction per pixel: see below.
//
// Bound: HBM (X read once per layer for dW / the forward, dX written once per layer: 33.5 MB each at batch 2 of cfg-2).
#include "common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxLayers = 48;
constexpr int kSeg = 32;             // pixels per unit of the weight-gradient kernel
constexpr int kRB = 3;               // output rows per unit of the weight-gradient kernel (kRB + 2 input rows in registers)
constexpr int kCh = 8;               // ... walked kCh pixels at a time
constexpr int kDSeg = 16;            // pixels per row of a unit of the data-gradient kernel (4 lanes-groups x 4 pixels)

struct ThinBArgs {
    const float *x[kMaxLayers];
    const float *dy[kMaxLayers];
    const float *w[kMaxLayers];
    float *dx[kMaxLayers];
    float *dw[kMaxLayers];
    float *db[kMaxLayers];
    unsigned char cout[kMaxLayers];
    unsigned char group[kMaxLayers]; // the layers of the launch's output-channel count (the kernels are instantiated per count)
    float *ws;
    int n, batch, in_h, in_w, out_h, out_w, pad, cin;
    int x_ld;                        // floats per pixel of the x / dx tensors (>= cin: the layers may read / write channel slices of wider maps)
    int segs, bands, units, wgs;     // weight gradient: 32-pixel segments per output row, bands of kRB rows, units per layer, workgroups per layer
    int dsegs, dunits;               // data gradient: 16-pixel segments per input row, rows x segments per layer (a unit is 1 or 2 rows)
    unsigned x_bytes;                // extent of one x / dx tensor
    unsigned ypix;                   // batch * out_h * out_w (dy of layer i holds ypix * cout[i] floats)
};

constexpr unsigned kOob = 0xc0000000u;     // byte offset beyond every buffer here (extents < 3 GiB are required), also after small immediates

// ------------------------------------------------------------------------------------------------ weight + bias gradient
template <int COUT>
__device__ __forceinline__ void thin_wgrad_body(const ThinBArgs &a, const int layer, float (*red)[10 * COUT][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave, waves = a.wgs * 4;
    const int ci = blockIdx.y * 64 + lane;
    const bool ci_ok = ci < a.cin;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x[layer], 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy[layer], 0, (int)(a.ypix * COUT * 4u), 0x00020000);
    float acc[10][COUT];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[t][c] = 0.f;
    const unsigned x_c = (unsigned)ci * 4u, xrow = (unsigned)a.x_ld * 4u;
    for (int u = gw; u < a.units; u += waves) {
        // a unit = kRB consecutive output rows of one 32-pixel segment: their kRB + 2 input rows are loaded once (a row alone would
        // pull its three input rows, i.e. every input element three times through L2)
        const int seg = u % a.segs, t0 = u / a.segs;
        const int band = t0 % a.bands, img = t0 / a.bands;
        const int oy0 = band * kRB;
        const int ox_begin = seg * kSeg, ox_end = min(ox_begin + kSeg, a.out_w);
        const int iy0 = oy0 - a.pad;
        unsigned rowoff[kRB + 2];
#pragma unroll
        for (int r = 0; r < kRB + 2; ++r) {
            const int iy = iy0 + r;
            rowoff[r] = (ci_ok && (unsigned)iy < (unsigned)a.in_h) ? (unsigned)((img * a.in_h + iy) * a.in_w) * xrow + x_c : kOob;
        }
        auto load_x = [&](int r, int ix) -> float {
            const unsigned off = (rowoff[r] != kOob && (unsigned)ix < (unsigned)a.in_w) ? rowoff[r] + (unsigned)ix * xrow : kOob;
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, 0, 0));
        };
        float w[kRB + 2][kCh + 2];
#pragma unroll
        for (int r = 0; r < kRB + 2; ++r) {
            w[r][0] = load_x(r, ox_begin - a.pad);
            w[r][1] = load_x(r, ox_begin - a.pad + 1);
        }
        for (int ox0 = ox_begin; ox0 < ox_end; ox0 += kCh) {
            // dY of kCh pixels of each row: lane l holds element l of the row's flat (pixel, channel) array starting at pixel ox0
            const int px = lane / COUT;
            const bool yok = ox0 + px < ox_end && px < kCh;    // pixels past the segment's end multiply dY = 0
            float dyv[kRB];
#pragma unroll
            for (int o = 0; o < kRB; ++o) {
                const int oy = oy0 + o;
                const unsigned ybase = (unsigned)((img * a.out_h + oy) * a.out_w) * (unsigned)(COUT * 4);
                dyv[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    y_rsrc, (yok && oy < a.out_h) ? ybase + (unsigned)(ox0 * COUT + lane) * 4u : kOob, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < kRB + 2; ++r)
#pragma unroll
                for (int j = 0; j < kCh; ++j) w[r][2 + j] = load_x(r, ox0 - a.pad + 2 + j);
#pragma unroll
            for (int o = 0; o < kRB; ++o)
#pragma unroll
                for (int j = 0; j < kCh; ++j) {
#pragma unroll
                    for (int c = 0; c < COUT; ++c) {
                        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyv[o]), j * COUT + c));
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int s2 = 0; s2 < 3; ++s2) acc[r * 3 + s2][c] = __builtin_fmaf(d, w[o + r][j + s2], acc[r * 3 + s2][c]);
                        acc[9][c] += d;                         // the bias gradient (the same sum in every lane)
                    }
                }
#pragma unroll
            for (int r = 0; r < kRB + 2; ++r) { w[r][0] = w[r][kCh]; w[r][1] = w[r][kCh + 1]; }
        }
    }
    // the four waves of the workgroup meet in LDS and are added in wave order; one partial set per workgroup
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) red[wave][t * COUT + c][lane] = acc[t][c];
    __syncthreads();
    float *ws = a.ws + ((size_t)layer * a.wgs + blockIdx.x) * (size_t)(10 * 4) * a.cin;
    for (int e = threadIdx.x; e < 10 * COUT * 64; e += 256) {
        const int l = e & 63, tc = e >> 6;
        const int cc = blockIdx.y * 64 + l;
        if (cc < a.cin) ws[(size_t)tc * a.cin + cc] = ((red[0][tc][l] + red[1][tc][l]) + red[2][tc][l]) + red[3][tc][l];
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinBArgs a) {
    __shared__ float red[4][10 * COUT][64];
    thin_wgrad_body<COUT>(a, a.group[blockIdx.z], red);
}

// dw[c][ci][tap] / db[c] = sum over the layer's workgroups (in order) of ws[layer][workgroup][tap (9 = bias)][c][ci]
__global__ __launch_bounds__(256) void thin_reduce_kernel(const ThinBArgs a) {
    const int layer = blockIdx.y, cout = a.cout[layer];
    const int total = 10 * cout * a.cin;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float *ws = a.ws + (size_t)layer * a.wgs * (size_t)(10 * 4) * a.cin + i;
    const size_t stride = (size_t)(10 * 4) * a.cin;
    float v = 0.f;
    int p = 0;
    for (; p + 8 <= a.wgs; p += 8) {                           // eight independent loads in flight, added in order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = ws[(size_t)(p + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    for (; p < a.wgs; ++p) v += ws[(size_t)p * stride];
    const int ci = i % a.cin, r = i / a.cin;
    const int c = r % cout, tap = r / cout;
    if (tap < 9) {
        if (a.dw[layer]) a.dw[layer][((size_t)c * a.cin + ci) * 9 + tap] = v;
    } else if (ci == 0 && a.db[layer]) {
        a.db[layer][c] = v;
    }
}

// ------------------------------------------------------------------------------------------------------- data gradient
template <int COUT>
__device__ __forceinline__ void thin_dgrad_body(const ThinBArgs &a, const int layer) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int ci = blockIdx.z * 64 + q * 4;
    const bool q_ok = ci < a.cin;                              // (cin is a multiple of 4)
    // this lane's weights W[c][ci .. ci + 3][r][s], kept for the whole launch
    f32x2 wlo[9][COUT], whi[9][COUT];
    {
        const float *w = a.w[layer];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = q_ok ? w[((size_t)c * a.cin + ci + k) * 9 + t] : 0.f;
                wlo[t][c] = f32x2{e[0], e[1]};
                whi[t][c] = f32x2{e[2], e[3]};
            }
    }
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy[layer], 0, (int)(a.ypix * COUT * 4u), 0x00020000);
    float *dx = a.dx[layer];
    const int stride_u = gridDim.x * 4;
    constexpr int kDR = COUT <= 3 ? 2 : 1;                     // rows per unit (4 output channels: the second row's window would spill)
    const int dbands = (a.in_h + kDR - 1) / kDR, dunits = a.batch * dbands * a.dsegs;
    for (int u = blockIdx.x * 4 + wave; u < dunits; u += stride_u) {
        // a unit = kDR consecutive rows of one 16-pixel segment: their kDR + 2 rows of dY are fetched once
        const int xs = u % a.dsegs, t0 = u / a.dsegs;
        const int band = t0 % dbands, img = t0 / dbands;
        const int iy0 = band * kDR;
        const int px0 = xs * kDSeg + g * 4;
        float d[kDR + 2][6][COUT];                             // dY rows iy0 + pad - 2 + q, columns px0 + pad - 2 + j
#pragma unroll
        for (int qq = 0; qq < kDR + 2; ++qq) {
            const int oy = iy0 + a.pad - 2 + qq;
            const bool row_ok = (unsigned)oy < (unsigned)a.out_h;
            const unsigned rowbase = (unsigned)((img * a.out_h + oy) * a.out_w);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ox = px0 + a.pad - 2 + j;
                const unsigned off = (row_ok && (unsigned)ox < (unsigned)a.out_w) ? (rowbase + (unsigned)ox) * (unsigned)(COUT * 4) : kOob;
#pragma unroll
                for (int c = 0; c < COUT; ++c)
                    d[qq][j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rsrc, off + (unsigned)c * 4u, 0, 0));
            }
        }
#pragma unroll
        for (int o = 0; o < kDR; ++o) {
            f32x2 lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = f32x2{0.f, 0.f}; hi[k] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int r = 0; r < 3; ++r)                        // output row iy0 + o, tap row r: dY row iy0 + o + pad - r = window row o + 2 - r
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                        for (int c = 0; c < COUT; ++c) {
                            const float dv = d[o + 2 - r][k + 2 - s2][c];
                            const f32x2 d2 = f32x2{dv, dv};
                            lo[k] = __builtin_elementwise_fma(d2, wlo[r * 3 + s2][c], lo[k]);
                            hi[k] = __builtin_elementwise_fma(d2, whi[r * 3 + s2][c], hi[k]);
                        }
            const int iy = iy0 + o;
            if (q_ok && iy < a.in_h) {
                float *row = dx + ((size_t)(img * a.in_h + iy) * a.in_w) * a.x_ld + ci;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (px0 + k < a.in_w) *reinterpret_cast<f32x4 *>(row + (size_t)(px0 + k) * a.x_ld) = f32x4{lo[k][0], lo[k][1], hi[k][0], hi[k][1]};
            }
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const ThinBArgs a) {
    thin_dgrad_body<COUT>(a, a.group[blockIdx.y]);
}

// ------------------------------------------------------------------------------------------------------------- forward
// y[p][c] = bias[c] + sum_{r,s,ci} X[p + (r, s) - pad][ci] W[c][ci][r][s] for cin <= 64: a lane owns 4 consecutive input channels of
// 4 consecutive pixels (16 lanes = the 64 channels of a pixel), its 9 x COUT x 4 weights in registers for the whole launch, the
// 3 x 6 window of float4 input values of a unit loaded once (coalesced 256-byte pixel rows), 9 x 4 x COUT FMAs per pixel and lane,
// then the 16 lanes of a pixel are added with four row-shift DPP steps.  f32 arithmetic in every mode.  The MFMA kernels pad
// these layers to 64 output columns and run them at 0.5 TB/s (64 us per 33 MB layer); here a layer is one pass at HBM speed.
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {          // value of lane (l - N) of the 16-lane row, 0 outside the row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int COUT>
__device__ __forceinline__ void thin_fwd_body(const ThinBArgs &a, const int layer) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int ci = q * 4;
    const bool q_ok = ci < a.cin;
    f32x4 wv[9][COUT];
    {
        const float *w = a.w[layer];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < COUT; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) wv[t][c][k] = q_ok ? w[((size_t)c * a.cin + ci + k) * 9 + t] : 0.f;
    }
    float bias[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) bias[c] = a.db[layer] ? a.db[layer][c] : 0.f;        // (forward: db carries the bias vector, dx the output)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x[layer], 0, (int)a.x_bytes, 0x00020000);
    float *y = a.dx[layer];
    const unsigned xrow = (unsigned)a.x_ld * 4u;
    const int osegs = (a.out_w + kDSeg - 1) / kDSeg, ounits = a.batch * a.out_h * osegs;
    const int stride_u = gridDim.x * 4;
    for (int u = blockIdx.x * 4 + wave; u < ounits; u += stride_u) {
        const int xs = u % osegs, t0 = u / osegs;
        const int oy = t0 % a.out_h, img = t0 / a.out_h;
        const int px0 = xs * kDSeg + g * 4;
        float acc[4][COUT];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[k][c] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy - a.pad + r;
            const bool row_ok = q_ok && (unsigned)iy < (unsigned)a.in_h;
            const unsigned rowbase = (unsigned)((img * a.in_h + iy) * a.in_w);
            f32x4 xv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ix = px0 - a.pad + j;
                const unsigned off = (row_ok && (unsigned)ix < (unsigned)a.in_w) ? (rowbase + (unsigned)ix) * xrow + (unsigned)ci * 4u : kOob;
                xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                    for (int c = 0; c < COUT; ++c) {
                        const f32x4 xx = xv[k + s2], ww = wv[r * 3 + s2][c];
                        acc[k][c] = __builtin_fmaf(xx[0], ww[0], __builtin_fmaf(xx[1], ww[1], __builtin_fmaf(xx[2], ww[2], __builtin_fmaf(xx[3], ww[3], acc[k][c]))));
                    }
        }
        // the 16 lanes of a pixel: lane 15 of the row ends up with the sum (fixed order)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                float v = acc[k][c];
                v += row_dpp<0x118>(v);          // row_shr:8
                v += row_dpp<0x114>(v);          // row_shr:4
                v += row_dpp<0x112>(v);          // row_shr:2
                v += row_dpp<0x111>(v);          // row_shr:1
                acc[k][c] = v;
            }
        if (q == 15) {
            float *row = y + ((size_t)(img * a.out_h + oy) * a.out_w) * COUT;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (px0 + k < a.out_w) {
#pragma unroll
                    for (int c = 0; c < COUT; ++c) row[(size_t)(px0 + k) * COUT + c] = acc[k][c] + bias[c];
                }
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const ThinBArgs a) {
    thin_fwd_body<COUT>(a, a.group[blockIdx.y]);
}

int fill(const sgv3d_conv_desc *d, int n, const int32_t *cout, ThinBArgs &a) {
    SGV3D_REQUIRE(d && cout, "conv3x3_thin_backward_batched: null descriptor / cout list");
    SGV3D_REQUIRE(n >= 1 && n <= kMaxLayers, "conv3x3_thin_backward_batched: 1 .. %d layers", kMaxLayers);
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad >= 0 && d->pad <= 2,
                  "conv3x3_thin_backward_batched: 3x3 / stride 1 / dilation 1 layers, pad 0 .. 2");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cin % 4 == 0 && d->x_ld >= d->cin && d->x_ld % 4 == 0 && d->x_coff == 0,
                  "conv3x3_thin_backward_batched: NHWC inputs with channel count and pixel stride multiples of 4 (x_ld >= cin: a channel slice "
                  "of a wider map is passed as the pointer to its first channel)");
    SGV3D_REQUIRE(d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2 && d->out_h > 0 && d->out_w > 0,
                  "conv3x3_thin_backward_batched: output size does not belong to this input size");
    const unsigned long long xb = ((unsigned long long)d->batch * d->in_h * d->in_w - 1) * d->x_ld * 4ull + d->cin * 4ull;      // extent from the slice's first channel
    const unsigned long long yp = (unsigned long long)d->batch * d->out_h * d->out_w;
    SGV3D_REQUIRE(xb < kOob && yp * 16ull < kOob, "conv3x3_thin_backward_batched: tensors (a slice: the extent from its first channel) must be smaller than 3 GiB");
    a = ThinBArgs{};
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(cout[i] >= 1 && cout[i] <= 4, "conv3x3_thin_backward_batched: 1 .. 4 output channels per layer (layer %d has %d)", i, cout[i]);
        a.cout[i] = (unsigned char)cout[i];
    }
    a.n = n; a.batch = d->batch; a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w; a.pad = d->pad; a.cin = d->cin; a.x_ld = d->x_ld;
    a.segs = cdiv(d->out_w, kSeg);
    a.bands = cdiv(d->out_h, kRB);
    a.units = d->batch * a.bands * a.segs;
    // about four waves per SIMD over the whole launch (1024 workgroups of 4 waves on 256 CUs), at least one unit per wave
    const int chunks = cdiv(d->cin, 64);
    // one launch per output-channel count: ~4 workgroups per CU for the SMALLEST of those launches (one count for all layers: the
    // partial sums of every layer have the same workspace stride)
    int cnt[5] = {0, 0, 0, 0, 0}, mmin = n;
    for (int i = 0; i < n; ++i) ++cnt[a.cout[i]];
    for (int c = 1; c <= 4; ++c)
        if (cnt[c] > 0 && cnt[c] < mmin) mmin = cnt[c];
    int wgs = cdiv(1024, mmin * chunks);
    wgs = wgs > 256 ? 256 : wgs;
    wgs = wgs < 1 ? 1 : wgs;
    wgs = wgs > cdiv(a.units, 4) ? cdiv(a.units, 4) : wgs;
    a.wgs = wgs;
    a.dsegs = cdiv(d->in_w, kDSeg);
    a.dunits = d->batch * d->in_h * a.dsegs;
    a.x_bytes = (unsigned)xb; a.ypix = (unsigned)yp;
    return SGV3D_OK;
}

}  // namespace

extern "C" size_t sgv3d_conv3x3_thin_backward_batched_workspace_bytes(const sgv3d_conv_desc *d, int n, const int32_t *cout) {
    ThinBArgs a;
    if (fill(d, n, cout, a) != SGV3D_OK) return 0;
    return (size_t)n * a.wgs * (10 * 4) * a.cin * sizeof(float);
}

extern "C" int sgv3d_conv3x3_thin_backward_batched(const sgv3d_conv_desc *d, int n, const int32_t *cout, const float *const *x_list,
                                                   const float *const *dy_list, const float *const *w_list, float *const *dx_list,
                                                   float *const *dw_list, float *const *db_list, void *workspace, size_t workspace_bytes,
                                                   void *stream) {
    ThinBArgs a;
    if (int rc = fill(d, n, cout, a)) return rc;
    SGV3D_REQUIRE(dy_list, "conv3x3_thin_backward_batched: null dy list");
    const bool want_w = dw_list != nullptr || db_list != nullptr, want_x = dx_list != nullptr;
    SGV3D_REQUIRE(!want_w || (x_list && workspace), "conv3x3_thin_backward_batched: weight / bias gradients need the inputs and the workspace");
    SGV3D_REQUIRE(!want_x || w_list, "conv3x3_thin_backward_batched: data gradients need the weights");
    const size_t need = (size_t)n * a.wgs * (10 * 4) * a.cin * sizeof(float);
    SGV3D_REQUIRE(!want_w || workspace_bytes >= need, "conv3x3_thin_backward_batched: workspace too small (%zu < %zu)", workspace_bytes, need);
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(dy_list[i] && (!want_w || x_list[i]) && (!want_x || (w_list[i] && dx_list[i])), "conv3x3_thin_backward_batched: null pointer (layer %d)", i);
        SGV3D_REQUIRE(!want_x || (reinterpret_cast<uintptr_t>(dx_list[i]) & 15) == 0, "conv3x3_thin_backward_batched: dx must be 16-byte aligned (layer %d)", i);
        a.dy[i] = dy_list[i];
        a.x[i] = want_w ? x_list[i] : nullptr;
        a.w[i] = want_x ? w_list[i] : nullptr;
        a.dx[i] = want_x ? dx_list[i] : nullptr;
        a.dw[i] = dw_list ? dw_list[i] : nullptr;
        a.db[i] = db_list ? db_list[i] : nullptr;
    }
    a.ws = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    const int chunks = cdiv(a.cin, 64);
    for (int c = 1; c <= 4; ++c) {                       // one launch per output-channel count present (CenterHead: 1, 2, 3)
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (a.cout[i] == c) a.group[m++] = (unsigned char)i;
        if (m == 0) continue;
        if (want_w) {
            const dim3 grid(a.wgs, chunks, m);
            switch (c) {
                case 1: thin_wgrad_kernel<1><<<grid, 256, 0, st>>>(a); break;
                case 2: thin_wgrad_kernel<2><<<grid, 256, 0, st>>>(a); break;
                case 3: thin_wgrad_kernel<3><<<grid, 256, 0, st>>>(a); break;
                default: thin_wgrad_kernel<4><<<grid, 256, 0, st>>>(a); break;
            }
            if (int rc = check_launch("thin_wgrad_kernel")) return rc;
        }
        if (want_x) {
            int gw = cdiv(1024, m * chunks);                // per launch, as above
            gw = gw > cdiv(a.dunits, 4) ? cdiv(a.dunits, 4) : gw;
            const dim3 grid(gw < 1 ? 1 : gw, m, chunks);
            switch (c) {
                case 1: thin_dgrad_kernel<1><<<grid, 256, 0, st>>>(a); break;
                case 2: thin_dgrad_kernel<2><<<grid, 256, 0, st>>>(a); break;
                case 3: thin_dgrad_kernel<3><<<grid, 256, 0, st>>>(a); break;
                default: thin_dgrad_kernel<4><<<grid, 256, 0, st>>>(a); break;
            }
            if (int rc = check_launch("thin_dgrad_kernel")) return rc;
        }
    }
    if (want_w) {
        thin_reduce_kernel<<<dim3(cdiv(10 * 4 * a.cin, 256), n), 256, 0, st>>>(a);
        if (int rc = check_launch("thin_reduce_kernel")) return rc;
    }
    return SGV3D_OK;
}

// Forward of the same n layers (cin <= 64): y_list[i] = conv3x3(x_list[i], w_list[i]) + bias_list[i], y CONTIGUOUS [batch, out_h, out_w, cout[i]]
// f32; bias_list or single entries of it may be NULL.  One launch per output-channel count present.
extern "C" int sgv3d_conv3x3_thin_forward_batched(const sgv3d_conv_desc *d, int n, const int32_t *cout, const float *const *x_list,
                                                  const float *const *w_list, const float *const *bias_list, float *const *y_list,
                                                  void *stream) {
    ThinBArgs a;
    if (int rc = fill(d, n, cout, a)) return rc;
    SGV3D_REQUIRE(a.cin <= 64, "conv3x3_thin_forward_batched: at most 64 input channels (got %d)", a.cin);
    SGV3D_REQUIRE(x_list && w_list && y_list, "conv3x3_thin_forward_batched: null list");
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(x_list[i] && w_list[i] && y_list[i], "conv3x3_thin_forward_batched: null pointer (layer %d)", i);
        SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(x_list[i]) & 15) == 0, "conv3x3_thin_forward_batched: x must be 16-byte aligned (layer %d)", i);
        a.x[i] = x_list[i];
        a.w[i] = w_list[i];
        a.dx[i] = y_list[i];                       // (the argument block's dx / db slots carry the output and the bias in the forward)
        a.db[i] = bias_list ? const_cast<float *>(bias_list[i]) : nullptr;
    }
    hipStream_t st = as_stream(stream);
    const int units = a.batch * a.out_h * cdiv(a.out_w, kDSeg);
    for (int c = 1; c <= 4; ++c) {
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (a.cout[i] == c) a.group[m++] = (unsigned char)i;
        if (m == 0) continue;
        int wgs = cdiv(1024, m);                           // ~4 workgroups per CU over THIS launch (the layers of one channel count)
        wgs = wgs > cdiv(units, 4) ? cdiv(units, 4) : wgs;
        const dim3 grid(wgs, m);
        switch (c) {
            case 1: thin_fwd_kernel<1><<<grid, 256, 0, st>>>(a); break;
            case 2: thin_fwd_kernel<2><<<grid, 256, 0, st>>>(a); break;
            case 3: thin_fwd_kernel<3><<<grid, 256, 0, st>>>(a); break;
            default: thin_fwd_kernel<4><<<grid, 256, 0, st>>>(a); break;
        }
        if (int rc = check_launch("thin_fwd_kernel")) return rc;
    }
    return SGV3D_OK;
}
