// Winograd F(4x4, 3x3) for gfx950 in three launches -- the form for layers with MANY channels on SMALL maps (HeightNet's ten
// 512 -> 512 convolutions at 54x96, ResNet layer 3 / 4, the 640-channel BEV trunk stage): 3x3 / stride 1 / pad 1 convolutions
// the reference runs through cuDNN (layers/backbones/lss_fpn.py:161-250 and the mmdet ResNet blocks it builds).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A     per 4x4 output tile (6x6 input tile, 36 positions)
//
// executes 36 / 16 / 9 = 1/4 of the direct form's multiplications, against 1/2.25 for the F(2x2) kernels of conv_wino.hip.
// With 36 positions the accumulators of a fused kernel no longer fit a wave (36 x 16 registers), so the three stages are
// separate launches and the middle one -- all of the matrix work -- is the f32 MFMA implicit-GEMM kernel itself, run as a
// GROUPED GEMM over the positions (conv_gemm_grouped, conv_igemm.hip):
//
//   1. wino4_input_kernel    x NHWC -> V[36][tiles_pad][cin]          (B^T d B; one thread per tile and 4 channels)
//   2. conv_gemm_grouped     M[p] = V[p] . U[p]^T, p = 0..35          (U[p][cout][cin] = (G g G^T)[p], packed per position;
//                                                                       SGV3D_TILE_48x64: conv_gemm_grouped16, gemm16_grouped.hip)
//   3. wino4_output_kernel   M[36][tiles_pad][cout] -> y NHWC         (A^T M A, folded BN / bias, residual, ReLU)
//
// V and M (28 MB each for a 512-channel 54x96 layer) are written once and read once; they stay in the 256 MB last-level
// cache between the launches.  The transforms are HBM / L2-bound streaming kernels (bound: HBM, 4 (1 + 2.25) bytes per input
// element and 4 (2.25 + 1 [+ 1]) per output element); the GEMM is bound by the f32 MFMA rate.  Pays where cin and cout are
// large (the transforms cost O(pixels x channels), the GEMM O(pixels x cin x cout)): the first-call measurement of hip_ops
// decides per layer.  fp32 rounding: the transformed operands are up to 100x the inputs (|B^T| row sums 10), so the error of an
// output is ~1e-5 of its scale against ~1e-6 for F(2x2); tests/test_conv_wino_gpu.py bounds it against float64.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

struct f4 {
    float x, y, z, w;
};
__device__ __forceinline__ f4 ld4(const float *p) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return {v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float *p, const f4 &v) { *reinterpret_cast<float4 *>(p) = make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f4 fma4(float s, const f4 &a, const f4 &b) {   // s * a + b
    return {fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w)};
}
__device__ __forceinline__ f4 add4(const f4 &a, const f4 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ f4 sub4(const f4 &a, const f4 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

// B^T (6 -> 6), 12 operations:
//   t0 = 4 d0 - 5 d2 + d4          t1 = (d4 - 4 d2) + (d3 - 4 d1)     t2 = (d4 - 4 d2) - (d3 - 4 d1)
//   t3 = (d4 - d2) + 2 (d3 - d1)   t4 = (d4 - d2) - 2 (d3 - d1)       t5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void wino4_bt(f4 &d0, f4 &d1, f4 &d2, f4 &d3, f4 &d4, f4 &d5) {
    const f4 a = fma4(-4.f, d2, d4), b = fma4(-4.f, d1, d3);
    const f4 c = sub4(d4, d2), e = sub4(d3, d1);
    const f4 t0 = fma4(4.f, d0, fma4(-5.f, d2, d4));
    const f4 t5 = fma4(4.f, d1, fma4(-5.f, d3, d5));
    d0 = t0;
    d1 = add4(a, b);
    d2 = sub4(a, b);
    d3 = fma4(2.f, e, c);
    d4 = fma4(-2.f, e, c);
    d5 = t5;
}

// A^T (6 -> 4), 10 operations:
//   y0 = m0 + (m1 + m2) + (m3 + m4)        y1 = (m1 - m2) + 2 (m3 - m4)
//   y2 = (m1 + m2) + 4 (m3 + m4)           y3 = (m1 - m2) + 8 (m3 - m4) + m5
__device__ __forceinline__ void wino4_at(const f4 &m0, const f4 &m1, const f4 &m2, const f4 &m3, const f4 &m4, const f4 &m5,
                                         f4 &y0, f4 &y1, f4 &y2, f4 &y3) {
    const f4 p = add4(m1, m2), q = sub4(m1, m2), r = add4(m3, m4), s = sub4(m3, m4);
    y0 = add4(add4(m0, p), r);
    y1 = fma4(2.f, s, q);
    y2 = fma4(4.f, r, p);
    y3 = add4(fma4(8.f, s, q), m5);
}

// Threads per workgroup of the transform kernels (single-wave workgroups -- 672 instead of 168 for a 512-channel 54x96 layer --
// measured the same in round 6: the transforms are bound by the latency of their 36 dependent-free loads and 36-108 stores per
// thread, not by how the waves spread over the CUs).
constexpr int kTB = 256;

struct Wino4Args {
    const float *x, *scale, *bias, *res;
    float *v, *m, *y;
    int batch, h, w, cin, cout, x_ld, x_coff, y_ld, y_coff, res_ld, relu;
    int ty, tx;            // tiles per image and phase
    int dil;               // dilation (pad == dil): the convolution splits into dil x dil independent ones on the sub-grids
                           // (py + dil * i, px + dil * j); a tile is 4x4 outputs / 6x6 inputs of ONE sub-grid
    int rows;              // rows of V / M per position (tiles of all images and phases, padded to a multiple of 64)
    int c0, cn;            // output-channel chunk [c0, c0 + cn) this pass of the GEMM / output transform covers (M holds cn columns)
    int gw;                // 0: NHWC output; > 0 (GROUP_PLANES): y is [cout / gw][pixels][gw], one NHWC map per group of gw channels
};

// tile index -> (image, sub-grid phase, tile position inside the sub-grid)
__device__ __forceinline__ void wino4_tile(const Wino4Args &a, int t, int &b, int &py, int &px, int &iy, int &ix) {
    const int per_phase = a.ty * a.tx, per_img = per_phase * a.dil * a.dil;
    b = t / per_img;
    int r = t - b * per_img;
    const int ph = r / per_phase;
    r -= ph * per_phase;
    py = ph / a.dil;
    px = ph - py * a.dil;
    iy = r / a.tx;
    ix = r - iy * a.tx;
}

// one thread: one tile x 4 input channels.  Consecutive threads = consecutive channel quads: 16-byte accesses, contiguous.
__global__ __launch_bounds__(kTB) void wino4_input_kernel(const Wino4Args a) {
    const int cq = a.cin >> 2;
    const long long i = (long long)blockIdx.x * kTB + threadIdx.x;
    const long long ntile = (long long)a.batch * a.ty * a.tx * a.dil * a.dil;
    if (i >= ntile * cq) return;
    const int t = (int)(i / cq), c = (int)(i - (long long)t * cq) * 4;
    int b, py, px, iy, ix;
    wino4_tile(a, t, b, py, px, iy, ix);
    const int y0 = py + (4 * iy - 1) * a.dil, x0 = px + (4 * ix - 1) * a.dil;
    const float *xb = a.x + (size_t)b * a.h * a.w * a.x_ld + a.x_coff + c;
    f4 d[6][6];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        const int yy = y0 + rr * a.dil;
        const bool rok = (unsigned)yy < (unsigned)a.h;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            const int xx = x0 + cc * a.dil;
            d[rr][cc] = (rok && (unsigned)xx < (unsigned)a.w) ? ld4(xb + ((size_t)yy * a.w + xx) * a.x_ld) : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) wino4_bt(d[0][cc], d[1][cc], d[2][cc], d[3][cc], d[4][cc], d[5][cc]);     // B^T d
    float *vb = a.v + (size_t)t * a.cin + c;
    const size_t plane = (size_t)a.rows * a.cin;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        wino4_bt(d[rr][0], d[rr][1], d[rr][2], d[rr][3], d[rr][4], d[rr][5]);                                 // (B^T d) B
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) st4(vb + (size_t)(rr * 6 + cc) * plane, d[rr][cc]);
    }
}

// one thread: one tile x 4 output channels
__global__ __launch_bounds__(kTB) void wino4_output_kernel(const Wino4Args a) {
    const int cq = a.cn >> 2;
    const long long i = (long long)blockIdx.x * kTB + threadIdx.x;
    const long long ntile = (long long)a.batch * a.ty * a.tx * a.dil * a.dil;
    if (i >= ntile * cq) return;
    const int t = (int)(i / cq), cl = (int)(i - (long long)t * cq) * 4;     // cl: column inside the chunk
    const int c = a.c0 + cl;                                                // output channel
    int b, py, px, iy, ix;
    wino4_tile(a, t, b, py, px, iy, ix);
    const float *mb = a.m + (size_t)t * a.cn + cl;
    const size_t plane = (size_t)a.rows * a.cn;
    f4 rr_[4][6];                                           // A^T M: 4 x 6
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
        const f4 m0 = ld4(mb + (size_t)(0 * 6 + cc) * plane), m1 = ld4(mb + (size_t)(1 * 6 + cc) * plane),
                 m2 = ld4(mb + (size_t)(2 * 6 + cc) * plane), m3 = ld4(mb + (size_t)(3 * 6 + cc) * plane),
                 m4 = ld4(mb + (size_t)(4 * 6 + cc) * plane), m5 = ld4(mb + (size_t)(5 * 6 + cc) * plane);
        wino4_at(m0, m1, m2, m3, m4, m5, rr_[0][cc], rr_[1][cc], rr_[2][cc], rr_[3][cc]);
    }
    const f4 sc = a.scale ? ld4(a.scale + c) : f4{1.f, 1.f, 1.f, 1.f};
    const f4 sh = a.bias ? ld4(a.bias + c) : f4{0.f, 0.f, 0.f, 0.f};
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        f4 o[4];
        wino4_at(rr_[u][0], rr_[u][1], rr_[u][2], rr_[u][3], rr_[u][4], rr_[u][5], o[0], o[1], o[2], o[3]);   // (A^T M) A
        const int yy = py + (4 * iy + u) * a.dil;
        if (yy >= a.h) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int xx = px + (4 * ix + v) * a.dil;
            if (xx >= a.w) continue;
            const size_t pix = ((size_t)b * a.h + yy) * a.w + xx;
            f4 q = {fmaf(o[v].x, sc.x, sh.x), fmaf(o[v].y, sc.y, sh.y), fmaf(o[v].z, sc.z, sh.z), fmaf(o[v].w, sc.w, sh.w)};
            if (a.res) q = add4(q, ld4(a.res + pix * a.res_ld + c));
            q = {fmaxf(q.x, floor_), fmaxf(q.y, floor_), fmaxf(q.z, floor_), fmaxf(q.w, floor_)};
            if (a.gw > 0) {            // plane of the channel's group (gw % 4 == 0: a quad never straddles groups)
                const int grp = c / a.gw;
                st4(a.y + ((size_t)grp * a.batch * a.h * a.w + pix) * a.gw + (c - grp * a.gw), q);
            } else {
                st4(a.y + pix * a.y_ld + a.y_coff + c, q);
            }
        }
    }
}

// U[p][co][ci] = (G g G^T)[i][j], p = 6 i + j, written straight into the 36 packed weight blocks [cout_pad][k_pad] of the
// grouped GEMM (k = ci: a 1x1 weight in either k order), zero rows / columns in the padding.  One thread per (co, ci).
//   G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]]
__global__ __launch_bounds__(256) void wino4_pack_weight_kernel(const float *__restrict__ w, int cout, int cin, int k_pad,
                                                                int cout_pad, float *__restrict__ u) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)cout_pad * k_pad) return;
    const int co = (int)(i / k_pad), ci = (int)(i - (long long)co * k_pad);
    float g[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (co < cout && ci < cin) {
        const float *p = w + ((size_t)co * cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = p[a * 3 + b];
    }
    // G g: 6 x 3 (rows of G applied to the columns of g), then (G g) G^T: 6 x 6
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float s02 = g[0][b] + g[2][b];
        t[0][b] = 0.25f * g[0][b];
        t[1][b] = (-1.f / 6.f) * (s02 + g[1][b]);
        t[2][b] = (-1.f / 6.f) * (s02 - g[1][b]);
        t[3][b] = (1.f / 24.f) * g[0][b] + (1.f / 12.f) * g[1][b] + (1.f / 6.f) * g[2][b];
        t[4][b] = (1.f / 24.f) * g[0][b] - (1.f / 12.f) * g[1][b] + (1.f / 6.f) * g[2][b];
        t[5][b] = g[2][b];
    }
    const size_t block = (size_t)cout_pad * k_pad;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float s02 = t[r][0] + t[r][2];
        const float o[6] = {0.25f * t[r][0], (-1.f / 6.f) * (s02 + t[r][1]), (-1.f / 6.f) * (s02 - t[r][1]),
                            (1.f / 24.f) * t[r][0] + (1.f / 12.f) * t[r][1] + (1.f / 6.f) * t[r][2],
                            (1.f / 24.f) * t[r][0] - (1.f / 12.f) * t[r][1] + (1.f / 6.f) * t[r][2], t[r][2]};
#pragma unroll
        for (int c = 0; c < 6; ++c) u[(size_t)(r * 6 + c) * block + i] = o[c];
    }
}

// ---- f32x3 forms (gemm_x3_grouped.hip): the operands of the position GEMMs as three bf16 planes ---------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// x = hi + mid + lo exactly (8 + 8 + 8 significand bits; the remainders x - hi and x - hi - mid are exact in f32)
__device__ __forceinline__ void split3(const f4 &x, bf16x4 &hi, bf16x4 &mid, bf16x4 &lo) {
    const f32x4v v = {x.x, x.y, x.z, x.w};
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4v r1 = v - __builtin_convertvector(hi, f32x4v);
    mid = __builtin_convertvector(r1, bf16x4);
    const f32x4v r2 = r1 - __builtin_convertvector(mid, f32x4v);
    lo = __builtin_convertvector(r2, bf16x4);
}

// as wino4_input_kernel, writing V3 [36][rows][cin/32][3][32] bf16: one thread = one tile x 4 input channels -> three 8-byte
// stores per position (the 8 threads of a 32-channel record fill its three 64-byte planes)
__global__ __launch_bounds__(kTB) void wino4_input_x3_kernel(const Wino4Args a) {
    const int cq = a.cin >> 2;
    const long long i = (long long)blockIdx.x * kTB + threadIdx.x;
    const long long ntile = (long long)a.batch * a.ty * a.tx * a.dil * a.dil;
    if (i >= ntile * cq) return;
    const int t = (int)(i / cq), c = (int)(i - (long long)t * cq) * 4;
    int b, py, px, iy, ix;
    wino4_tile(a, t, b, py, px, iy, ix);
    const int y0 = py + (4 * iy - 1) * a.dil, x0 = px + (4 * ix - 1) * a.dil;
    const float *xb = a.x + (size_t)b * a.h * a.w * a.x_ld + a.x_coff + c;
    f4 d[6][6];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        const int yy = y0 + rr * a.dil;
        const bool rok = (unsigned)yy < (unsigned)a.h;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            const int xx = x0 + cc * a.dil;
            d[rr][cc] = (rok && (unsigned)xx < (unsigned)a.w) ? ld4(xb + ((size_t)yy * a.w + xx) * a.x_ld) : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) wino4_bt(d[0][cc], d[1][cc], d[2][cc], d[3][cc], d[4][cc], d[5][cc]);     // B^T d
    // record (row t, k-chunk c / 32) of position p: bf16 index ((p * rows + t) * (cin / 32) + c / 32) * 96 + plane * 32 + c % 32
    __bf16 *vb = reinterpret_cast<__bf16 *>(a.v) + ((size_t)t * (a.cin >> 5) + (c >> 5)) * 96 + (c & 31);
    const size_t plane = (size_t)a.rows * (a.cin >> 5) * 96;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        wino4_bt(d[rr][0], d[rr][1], d[rr][2], d[rr][3], d[rr][4], d[rr][5]);                                 // (B^T d) B
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            bf16x4 hi, mid, lo;
            split3(d[rr][cc], hi, mid, lo);
            __bf16 *o = vb + (size_t)(rr * 6 + cc) * plane;
            *reinterpret_cast<bf16x4 *>(o) = hi;
            *reinterpret_cast<bf16x4 *>(o + 32) = mid;
            *reinterpret_cast<bf16x4 *>(o + 64) = lo;
        }
    }
}

// U3[p][co / 16][ci / 32][plane][fragment order] = the three bf16 terms of (G g G^T)[i][j], p = 6 i + j; zero rows / columns in the padding.
// One thread per (co, ci); the arithmetic of wino4_pack_weight_kernel, then the split.
__global__ __launch_bounds__(256) void wino4_pack_weight_x3_kernel(const float *__restrict__ w, int cout, int cin, int k_pad,
                                                                   int cout_pad, __bf16 *__restrict__ u) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)cout_pad * k_pad) return;
    const int co = (int)(i / k_pad), ci = (int)(i - (long long)co * k_pad);
    float g[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (co < cout && ci < cin) {
        const float *p = w + ((size_t)co * cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = p[a * 3 + b];
    }
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float s02 = g[0][b] + g[2][b];
        t[0][b] = 0.25f * g[0][b];
        t[1][b] = (-1.f / 6.f) * (s02 + g[1][b]);
        t[2][b] = (-1.f / 6.f) * (s02 - g[1][b]);
        t[3][b] = (1.f / 24.f) * g[0][b] + (1.f / 12.f) * g[1][b] + (1.f / 6.f) * g[2][b];
        t[4][b] = (1.f / 24.f) * g[0][b] - (1.f / 12.f) * g[1][b] + (1.f / 6.f) * g[2][b];
        t[5][b] = g[2][b];
    }
    const size_t block = (size_t)cout_pad * k_pad * 3;
    // fragment order (gemm_x3_grouped.hip): block of 16 output channels x k-step x plane = 512 elements, (channel c, k) at
    // ((k / 8) * 16 + c) * 8 + k % 8
    __bf16 *ub = u + (((size_t)(co >> 4) * (k_pad >> 5) + (ci >> 5)) * 3) * 512 + ((((ci & 31) >> 3) * 16 + (co & 15)) * 8 + (ci & 7));
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float s02 = t[r][0] + t[r][2];
        const float o[6] = {0.25f * t[r][0], (-1.f / 6.f) * (s02 + t[r][1]), (-1.f / 6.f) * (s02 - t[r][1]),
                            (1.f / 24.f) * t[r][0] + (1.f / 12.f) * t[r][1] + (1.f / 6.f) * t[r][2],
                            (1.f / 24.f) * t[r][0] - (1.f / 12.f) * t[r][1] + (1.f / 6.f) * t[r][2], t[r][2]};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float v = o[c];
            const __bf16 hi = (__bf16)v;
            const float r1 = v - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)mid);
            __bf16 *q = ub + (size_t)(r * 6 + c) * block;
            q[0] = hi;
            q[512] = mid;
            q[1024] = lo;
        }
    }
}

// tiles of one sub-grid (the largest one: phase 0), of all phases and images, and the padded row count per position
void wino4_geom(const sgv3d_conv_desc *d, int &ty, int &tx, long long &tiles, int &rows) {
    const int dil = d->dil;
    ty = cdiv(cdiv(d->out_h, dil), 4);
    tx = cdiv(cdiv(d->out_w, dil), 4);
    tiles = (long long)d->batch * dil * dil * ty * tx;
    const int g = (d->tile & SGV3D_TILE_X3) ? gemm_x3_tile_rows(d->tile & 15)
                  : (d->tile & ~SGV3D_TILE_OCC5) == SGV3D_TILE_32x128 ? 32 : d->tile == SGV3D_TILE_48x64 ? 48 : 64;   // the GEMM's m-tile height
    rows = (int)((tiles + g - 1) / g * g);
}

// Output channels per pass: M of a pass (36 x rows x chunk floats) is kept below ~160 MB so that it is still in the 256 MB
// last-level cache when the output transform reads it back -- the 64 -> 2304 first layers of the CenterHead branches at
// 256x256 would otherwise write and re-read 1.36 GB.  Multiples of 128 (the packed weight blocks' row granularity).
int wino4_chunk(const sgv3d_conv_desc *d, int rows) {
    const long long per_channel = 36LL * rows * 4;
    long long cn = (160LL << 20) / per_channel / 128 * 128;
    if (cn < 128) cn = 128;
    return cn >= d->cout ? d->cout : (int)cn;
}

}  // namespace

// w_src: OIHW [cout, cin, 3, 3] f32 (cin may be smaller than the layer's padded channel count: the rest is zero) -> u_packed:
// 36 x cout_pad x k_pad floats, (cout_pad, k_pad) = sgv3d_conv_pack_geometry(cin_pad, cout)
extern "C" int sgv3d_conv_winograd4_pack_weight(const float *w_src, int cout, int cin, int k_pad, int cout_pad, float *u_packed,
                                                void *stream) {
    SGV3D_REQUIRE(w_src && u_packed && cout > 0 && cin > 0 && k_pad >= cin && k_pad % 32 == 0 && cout_pad >= cout,
                  "conv_winograd4_pack_weight: bad arguments (cout=%d cin=%d k_pad=%d cout_pad=%d)", cout, cin, k_pad, cout_pad);
    const long long total = (long long)cout_pad * k_pad;
    hipLaunchKernelGGL(wino4_pack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, cout, cin, k_pad,
                       cout_pad, u_packed);
    return check_launch("wino4_pack_weight_kernel");
}

// ... as three bf16 planes per element for the f32x3 position GEMM (desc.tile = SGV3D_TILE_X3 | variant): u3_packed is
// 36 x cout_pad x cin_pad x 3 bf16 (cin_pad % 32 == 0, cout_pad = cout rounded up to 32), layout [p][co][ci / 32][plane][ci % 32]
extern "C" int sgv3d_conv_winograd4_pack_weight_x3(const float *w_src, int cout, int cin, int cin_pad, int cout_pad, void *u3_packed,
                                                   void *stream) {
    SGV3D_REQUIRE(w_src && u3_packed && cout > 0 && cin > 0 && cin_pad >= cin && cin_pad % 32 == 0 && cout_pad >= cout && cout_pad % 32 == 0,
                  "conv_winograd4_pack_weight_x3: bad arguments (cout=%d cin=%d cin_pad=%d cout_pad=%d)", cout, cin, cin_pad, cout_pad);
    const long long total = (long long)cout_pad * cin_pad;
    hipLaunchKernelGGL(wino4_pack_weight_x3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, cout, cin, cin_pad,
                       cout_pad, static_cast<__bf16 *>(u3_packed));
    return check_launch("wino4_pack_weight_x3_kernel");
}

// bytes of V + M
extern "C" size_t sgv3d_conv2d_winograd4_workspace_bytes(const sgv3d_conv_desc *d) {
    if (!d || d->batch <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->cin <= 0 || d->cout <= 0 || d->dil <= 0) return 0;
    int ty, tx, rows;
    long long tiles;
    wino4_geom(d, ty, tx, tiles, rows);
    if (d->tile & SGV3D_TILE_X3)         // V as three bf16 planes (6 bytes per element), M f32
        return 36 * (size_t)rows * (6 * (size_t)d->cin + sizeof(float) * (size_t)wino4_chunk(d, rows));
    return sizeof(float) * 36 * (size_t)rows * ((size_t)d->cin + (size_t)wino4_chunk(d, rows));
}

// u_packed: 36 blocks [cout_pad][k_pad] (sgv3d_conv_pack_geometry(cin, cout)), block p = i * 6 + j the 1x1 weight
// (G g G^T)[i][j] packed by sgv3d_conv_pack_weight.  desc: as for sgv3d_conv2d_winograd_forward (NORMAL mode, no gate, no
// split-K); desc.k_pad / desc.cout_pad describe ONE block; desc.tile: SGV3D_TILE_64x64 (default) or SGV3D_TILE_64x128 for
// the grouped GEMM.
extern "C" int sgv3d_conv2d_winograd4_forward(const sgv3d_conv_desc *d, const float *x, const float *u_packed,
                                              const float *scale, const float *bias, const float *residual, float *y,
                                              void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(d && x && u_packed && y && workspace, "conv2d_winograd4_forward: null pointer");
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil >= 1 && d->pad == d->dil,
                  "conv2d_winograd4_forward: only 3x3 / stride 1 / pad == dilation");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0 && d->out_h == d->in_h && d->out_w == d->in_w,
                  "conv2d_winograd4_forward: bad sizes");
    const bool planes = d->mode == SGV3D_CONV_GROUP_PLANES;
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL || planes, "conv2d_winograd4_forward: NHWC or GROUP_PLANES output only");
    SGV3D_REQUIRE(!planes || (d->deconv_ks > 0 && d->deconv_ks % 4 == 0 && d->cout % d->deconv_ks == 0 && residual == nullptr),
                  "conv2d_winograd4_forward: GROUP_PLANES needs deconv_ks = group width (a multiple of 4) dividing cout, no residual");
    SGV3D_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && (d->x_ld & 3) == 0 && (d->x_coff & 3) == 0 &&
                      (planes || ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0)) && (residual == nullptr || (d->res_ld & 3) == 0),
                  "conv2d_winograd4_forward: channel counts, leading dimensions and offsets must be multiples of 4");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(u_packed) |
                    reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                  "conv2d_winograd4_forward: pointers must be 16-B aligned");
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin && (planes || d->y_ld >= d->y_coff + d->cout) && (residual == nullptr || d->res_ld >= d->cout),
                  "conv2d_winograd4_forward: leading dimension too small");
    const size_t need = sgv3d_conv2d_winograd4_workspace_bytes(d);
    if (workspace_bytes < need) return fail(SGV3D_ENOSPACE, "conv2d_winograd4_forward: workspace has %zu bytes, needs %zu", workspace_bytes, need);
    Wino4Args a;
    a.x = x; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.batch = d->batch; a.h = d->in_h; a.w = d->in_w; a.cin = d->cin; a.cout = d->cout;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld; a.relu = d->relu;
    long long tiles;
    wino4_geom(d, a.ty, a.tx, tiles, a.rows);
    a.dil = d->dil;
    SGV3D_REQUIRE(tiles < 0x7fffffffLL / 64, "conv2d_winograd4_forward: too many tiles");
    const bool x3 = (d->tile & SGV3D_TILE_X3) != 0;
    a.v = static_cast<float *>(workspace);
    a.m = x3 ? reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + (size_t)36 * a.rows * d->cin * 6)
             : a.v + (size_t)36 * a.rows * d->cin;
    hipStream_t st = as_stream(stream);
    a.gw = planes ? d->deconv_ks : 0;
    a.c0 = 0; a.cn = d->cout;
    if (x3) {
        // f32x3: u_packed is the three-plane bf16 form (sgv3d_conv_winograd4_pack_weight_x3), desc.cout_pad its rows per block
        SGV3D_REQUIRE(d->cin % 32 == 0 && d->cout_pad % 32 == 0 && d->cout_pad >= d->cout,
                      "conv2d_winograd4_forward: SGV3D_TILE_X3 needs cin %% 32 == 0 and cout_pad (a multiple of 32) of the x3 weights");
        SGV3D_REQUIRE((d->tile & 15) < 10, "conv2d_winograd4_forward: unknown SGV3D_TILE_X3 variant %d", d->tile & 15);
        hipLaunchKernelGGL(wino4_input_x3_kernel, dim3(cdiv(tiles * (d->cin / 4), kTB)), dim3(kTB), 0, st, a);
        const int chunk3 = wino4_chunk(d, a.rows);
        for (int c0 = 0; c0 < d->cout; c0 += chunk3) {
            a.c0 = c0;
            a.cn = d->cout - c0 < chunk3 ? d->cout - c0 : chunk3;
            // block p of the chunk's weights: cout_pad rows of cin * 6 bytes after block p - 1, shifted by c0 rows
            if (int rc = conv_gemm_grouped_x3(a.v, static_cast<const unsigned char *>(static_cast<const void *>(u_packed)) + (size_t)c0 * d->cin * 6,
                                              a.m, a.rows, d->cin, a.cn, d->cout_pad - c0, d->tile & 15, st, (size_t)d->cout_pad * d->cin * 6)) return rc;
            hipLaunchKernelGGL(wino4_output_kernel, dim3(cdiv(tiles * (a.cn / 4), kTB)), dim3(kTB), 0, st, a);
        }
        return check_launch("conv2d_winograd4_forward(x3)");
    }
    hipLaunchKernelGGL(wino4_input_kernel, dim3(cdiv(tiles * (d->cin / 4), kTB)), dim3(kTB), 0, st, a);
    int tile = (d->tile == SGV3D_TILE_64x128 || d->tile == SGV3D_TILE_32x128 || d->tile == SGV3D_TILE_48x64) ? d->tile
               : (d->tile & SGV3D_TILE_OCC5) ? (SGV3D_TILE_64x64 | SGV3D_TILE_OCC5) : SGV3D_TILE_64x64;
    if (tile == SGV3D_TILE_32x128 && (d->k_order != 1 || d->cin < 128)) tile = SGV3D_TILE_64x64;    // (what the narrow tile covers)
    SGV3D_REQUIRE(tile != SGV3D_TILE_48x64 || (d->k_order == 1 && d->cin % 32 == 0),
                  "conv2d_winograd4_forward: SGV3D_TILE_48x64 needs channel-chunk-major weights (cin %% 32 == 0)");
    // one pass per chunk of output channels: block p of the chunk's weights is cout_pad x k_pad floats after block p - 1
    // like the full blocks, shifted by c0 rows
    const int chunk = wino4_chunk(d, a.rows);
    for (int c0 = 0; c0 < d->cout; c0 += chunk) {
        a.c0 = c0;
        a.cn = d->cout - c0 < chunk ? d->cout - c0 : chunk;
        if (tile == SGV3D_TILE_48x64) {
            if (int rc = conv_gemm_grouped16(a.v, u_packed + (size_t)c0 * d->k_pad, a.m, a.rows, 36, d->cin, a.cn, d->k_pad,
                                             d->cout_pad, st)) return rc;
        } else if (int rc = conv_gemm_grouped(a.v, u_packed + (size_t)c0 * d->k_pad, a.m, a.rows, 36, d->cin, a.cn, d->k_pad,
                                              d->cout_pad, d->k_order, tile, st)) return rc;
        hipLaunchKernelGGL(wino4_output_kernel, dim3(cdiv(tiles * (a.cn / 4), kTB)), dim3(kTB), 0, st, a);
    }
    return check_launch("conv2d_winograd4_forward");
}
