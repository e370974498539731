// Small layers around the convolutions (all NHWC fp32, all HBM/latency-bound, no MFMA).
// Reference call sites are cited per kernel; paths relative to the reference repo.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kBlock = 256;

// nn.MaxPool2d(3, stride 2, pad 1) of the mmdet ResNet stem (built at layers/backbones/lss_fpn.py:296)
__global__ __launch_bounds__(kBlock) void maxpool3x3s2_kernel(int B, int H, int W, int C4, int OH, int OW,
                                                              const float4 *__restrict__ x, float4 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long total = (long long)B * OH * OW * C4;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh * 2 - 1 + dy;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iw = ow * 2 - 1 + dx;
            if ((unsigned)iw >= (unsigned)W) continue;
            const float4 v = x[((long long)(b * H + ih) * W + iw) * C4 + c];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    y[i] = m;
}

// the same pooling on a bf16 map (bf16-activation mode): 8 channels per thread, max is exact in bf16
typedef __bf16 mp_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(kBlock) void maxpool3x3s2_bf16_kernel(int B, int H, int W, int C8, int OH, int OW,
                                                                   const mp_bf16x8 *__restrict__ x, mp_bf16x8 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long total = (long long)B * OH * OW * C8;
    if (i >= total) return;
    const int c = (int)(i % C8);
    long long t = i / C8;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh * 2 - 1 + dy;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iw = ow * 2 - 1 + dx;
            if ((unsigned)iw >= (unsigned)W) continue;
            const mp_bf16x8 v = x[((long long)(b * H + ih) * W + iw) * C8 + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
        }
    }
    mp_bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)m[j];
    y[i] = o;
}

// image ingest: NCHW -> NHWC with zero-padded channels (the harness hands [B,1,1,3,H,W], exps/...:246)
__global__ __launch_bounds__(kBlock) void nchw_to_nhwc_kernel(int B, int C, long long HW, int Cp,
                                                              const float *__restrict__ x, float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;  // one pixel per thread
    if (i >= (long long)B * HW) return;
    const long long b = i / HW, p = i - b * HW;
    const float *src = x + b * C * HW + p;
    float *dst = y + i * Cp;
    for (int c = 0; c < Cp; ++c) dst[c] = c < C ? src[(long long)c * HW] : 0.f;
}

// NHWC channel slice -> NCHW through a 32x32 LDS transpose (coalesced on both sides)
__global__ __launch_bounds__(kBlock) void nhwc_to_nchw_kernel(int C, long long HW, int ld, int coff,
                                                              const float *__restrict__ x, float *__restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? x[((long long)b * HW + p) * ld + coff + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long p = p0 + tx;
        if (c < C && p < HW) y[((long long)b * C + c) * HW + p] = tile[tx][r];
    }
}

// nn.AdaptiveAvgPool2d((1,1))  (ASPP.global_avg_pool, layers/backbones/lss_fpn.py:81-86)
// Two deterministic stages: kAvgChunks workgroups per (image, 64-channel group) sum a pixel range each
// into the caller's workspace, then one wave per output adds the partials in fixed order.
constexpr int kAvgChunks = 32;

__global__ __launch_bounds__(kBlock) void global_avgpool_partial_kernel(int P, int C, int ld,
                                                                        const float *__restrict__ x,
                                                                        float *__restrict__ part) {
    __shared__ float red[4][64];
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int per = (P + kAvgChunks - 1) / kAvgChunks;
    const int p0 = chunk * per, p1 = min(P, p0 + per);
    float s = 0.f;
    if (c < C)
        for (int p = p0 + g; p < p1; p += 4) s += x[((long long)b * P + p) * ld + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C)
        part[((long long)b * kAvgChunks + chunk) * C + c] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void global_avgpool_final_kernel(int B, int P, int C,
                                                                      const float *__restrict__ part,
                                                                      float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;   // b*C + c
    if (i >= (long long)B * C) return;
    const int b = (int)(i / C), c = (int)(i - (long long)b * C);
    float s = 0.f;
    for (int k = 0; k < kAvgChunks; ++k) s += part[((long long)b * kAvgChunks + k) * C + c];
    y[i] = s / (float)P;
}

// y[b,n] = act(scale[n] * dot(W[n,:], x[b,:]) + bias[n]) : one wave per output
// (Mlp fc1/fc2, SELayer conv_reduce/conv_expand on [B,C,1,1], ASPP pooled 1x1; lss_fpn.py:122-159,81-86)
__global__ __launch_bounds__(kBlock) void dense_kernel(int B, int K, int N, const float *__restrict__ x,
                                                       const float *__restrict__ w, const float *__restrict__ scale,
                                                       const float *__restrict__ bias, int act, float *__restrict__ y,
                                                       const int *__restrict__ run) {
    if (run && *run == 0) return;                 // (sgv3d_dense_gated: y keeps what an earlier launch wrote)
    const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= B * N) return;
    const int b = wave / N, n = wave - b * N;
    const float *xr = x + (long long)b * K;
    const float *wr = w + (long long)n * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += wr[k] * xr[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        float v = s * (scale ? scale[n] : 1.f) + (bias ? bias[n] : 0.f);
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = 1.f / (1.f + expf(-v));
        y[(long long)b * N + n] = v;
    }
}

// F.interpolate of a 1x1 map to HxW (bilinear, align_corners) == broadcast  (lss_fpn.py:101-104)
__global__ __launch_bounds__(kBlock) void broadcast_channels_kernel(long long total, int P, int C, int ld, int coff,
                                                                    const float *__restrict__ v, float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long bp = i / C;
    const long long b = bp / P;
    y[bp * ld + coff + c] = v[b * C + c];
}

// SELayer gate (lss_fpn.py:155-159): y = x * gate[b, c]
__global__ __launch_bounds__(kBlock) void scale_channels_kernel(long long total4, int P, int C4,
                                                                const float4 *__restrict__ x,
                                                                const float4 *__restrict__ gate,
                                                                float4 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % C4);
    const long long b = i / ((long long)P * C4);
    const float4 v = x[i], g = gate[b * C4 + c];
    y[i] = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
}

__global__ __launch_bounds__(kBlock) void copy_channels_kernel(long long total, int C, int ld, int coff,
                                                               const float *__restrict__ x, float *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const long long bp = i / C;
    const int c = (int)(i - bp * C);
    y[i] = x[bp * ld + coff + c];
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)  (TaskFPN, bsm_lss_fpn.py:210)
__global__ __launch_bounds__(kBlock) void upsample_bilinear2x_kernel(int B, int H, int W, int C4,
                                                                     const float4 *__restrict__ x, float4 *__restrict__ y) {
    const long long total = (long long)B * 2 * H * 2 * W * C4;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ow = (int)(t % (2 * W));
    t /= (2 * W);
    const int oh = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    // area_pixel_compute_source_index(scale = 0.5, align_corners = false): max(0.5*(dst+0.5)-0.5, 0)
    const float sh = fmaxf(0.5f * ((float)oh + 0.5f) - 0.5f, 0.f), sw = fmaxf(0.5f * ((float)ow + 0.5f) - 0.5f, 0.f);
    const int h1 = (int)sh, w1 = (int)sw;
    const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
    const float lh1 = sh - (float)h1, lw1 = sw - (float)w1;
    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const float4 *p = x + ((long long)b * H * W) * C4 + c;
    const float4 a = p[((long long)h1 * W + w1) * C4], bq = p[((long long)h1 * W + w1 + w1p) * C4];
    const float4 cq = p[((long long)(h1 + h1p) * W + w1) * C4], d = p[((long long)(h1 + h1p) * W + w1 + w1p) * C4];
    float4 o;
    o.x = lh0 * (lw0 * a.x + lw1 * bq.x) + lh1 * (lw0 * cq.x + lw1 * d.x);
    o.y = lh0 * (lw0 * a.y + lw1 * bq.y) + lh1 * (lw0 * cq.y + lw1 * d.y);
    o.z = lh0 * (lw0 * a.z + lw1 * bq.z) + lh1 * (lw0 * cq.z + lw1 * d.z);
    o.w = lh0 * (lw0 * a.w + lw1 * bq.w) + lh1 * (lw0 * cq.w + lw1 * d.w);
    y[i] = o;
}

// Scalar-channel forward for maps whose channel count is not a multiple of 4 (the 7-class semantic logits).
__global__ void __launch_bounds__(kBlock) upsample2x_scalar_kernel(int B, int H, int W, int C, const float *__restrict__ x,
                                                               float *__restrict__ y) {
    const long long total = (long long)B * 4 * H * W * C;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long t = i / C;
    const int ow = (int)(t % (2 * W));
    t /= (2 * W);
    const int oh = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const float sh = fmaxf(0.5f * ((float)oh + 0.5f) - 0.5f, 0.f), sw = fmaxf(0.5f * ((float)ow + 0.5f) - 0.5f, 0.f);
    const int h1 = (int)sh, w1 = (int)sw;
    const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
    const float lh1 = sh - (float)h1, lw1 = sw - (float)w1;
    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const float *p = x + (long long)b * H * W * C + c;
    const float a = p[((long long)h1 * W + w1) * C], bq = p[((long long)h1 * W + w1 + w1p) * C];
    const float cq = p[((long long)(h1 + h1p) * W + w1) * C], d = p[((long long)(h1 + h1p) * W + w1 + w1p) * C];
    y[i] = lh0 * (lw0 * a + lw1 * bq) + lh1 * (lw0 * cq + lw1 * d);
}

// SABlock + residual (bsm_lss_fpn.py:151-160, 211): y = a + b * sigmoid(c)
__global__ __launch_bounds__(kBlock) void add_mul_sigmoid_kernel(long long n4, const float4 *__restrict__ a,
                                                                 const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                                 float4 *__restrict__ y) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n4) return;
    const float4 av = a[i], bv = b[i], cv = c[i];
    float4 o;
    o.x = av.x + bv.x * (1.f / (1.f + expf(-cv.x)));
    o.y = av.y + bv.y * (1.f / (1.f + expf(-cv.y)));
    o.z = av.z + bv.z * (1.f / (1.f + expf(-cv.z)));
    o.w = av.w + bv.w * (1.f / (1.f + expf(-cv.w)));
    y[i] = o;
}

// Background suppression of BSMLSSFPN (bsm_lss_fpn.py:524-529), in place on the [B,P,ld] buffer that
// already holds depth logits [0,D) and context [D,D+ctx): semantic = softmax(sem_logits); the
// transferred feature is cat(context, semantic) zeroed where semantic[0] (background) > thr.
__global__ __launch_bounds__(kBlock) void bsm_compose_kernel(long long pixels, int D, int ctx, int sem, int ld,
                                                             const float *__restrict__ sem_logits, int sem_ld, float thr,
                                                             float *__restrict__ buf) {
    const long long p = (long long)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);   // 32 lanes per pixel
    const int l = threadIdx.x & 31;
    if (p >= pixels) return;
    const float *sl = sem_logits + p * sem_ld;
    float m = -INFINITY;
    for (int k = 0; k < sem; ++k) m = fmaxf(m, sl[k]);
    float den = 0.f;
    for (int k = 0; k < sem; ++k) den += expf(sl[k] - m);
    const float p0 = expf(sl[0] - m) / den;
    const float keep = p0 > thr ? 0.f : 1.f;          // tran_feat * (1 - mask.int())
    float *row = buf + p * ld + D;
    for (int c = l; c < ctx; c += 32) row[c] = row[c] * keep;
    for (int k = l; k < ld - D - ctx; k += 32) row[ctx + k] = k < sem ? expf(sl[k] - m) / den * keep : 0.f;
}

// mmcv DeformConv2dPack sampling (DCNv1: deformable_im2col + bilinear with zero padding;
// configured at lss_fpn.py:190-198: 3x3, pad 1, stride 1, dil 1, deform_groups 1)
// One half-wave (32 lanes) per (pixel, tap): the sampling position, the four corner offsets and the bilinear weights are
// computed once (32-bit arithmetic) and the lanes walk over the channel quads -- the first version recomputed them, with
// 64-bit divisions, for every quad (this kernel runs between f32-MFMA convolutions and every vector instruction of it is
// MFMA time of the SIMD it runs on, tools/valu_tax.py).
__global__ __launch_bounds__(kBlock) void deform_im2col3x3_kernel(int B, int H, int W, int C, int groups,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ off, int off_ld,
                                                                  float *__restrict__ col) {
    const int C4 = C >> 2;
    const int lane = threadIdx.x & 31;
    const unsigned pt = blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);      // (pixel, tap) pair
    const unsigned npt = (unsigned)B * H * W * 9;
    if (pt >= npt) return;
    const unsigned t = pt / 9u;                 // pixel index b*H*W + h*W + w
    const int tap = (int)(pt - t * 9u);
    const unsigned t2 = t / (unsigned)W;
    const int w_ = (int)(t - t2 * W);
    const int b = (int)(t2 / (unsigned)H);
    const int h_ = (int)(t2 - (unsigned)b * H);
    const int ky = tap / 3, kx = tap - ky * 3;
    const float oy = off[(size_t)t * off_ld + 2 * tap], ox = off[(size_t)t * off_ld + 2 * tap + 1];
    const float hf = (float)(h_ - 1 + ky) + oy;
    const float wf = (float)(w_ - 1 + kx) + ox;
    const bool inside = hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W;
    const int h_low = (int)floorf(hf), w_low = (int)floorf(wf);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = hf - (float)h_low, lw = wf - (float)w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    // a corner outside the image contributes zero: weight 0 and a clamped (valid) address
    const bool ok1 = inside && h_low >= 0 && w_low >= 0, ok2 = inside && h_low >= 0 && w_high <= W - 1;
    const bool ok3 = inside && h_high <= H - 1 && w_low >= 0, ok4 = inside && h_high <= H - 1 && w_high <= W - 1;
    const float w1 = ok1 ? hh * hw : 0.f, w2 = ok2 ? hh * lw : 0.f, w3 = ok3 ? lh * hw : 0.f, w4 = ok4 ? lh * lw : 0.f;
    const float *base = x + (size_t)b * H * W * C;
    const float *p1 = base + (ok1 ? ((size_t)h_low * W + w_low) * C : 0);
    const float *p2 = base + (ok2 ? ((size_t)h_low * W + w_high) * C : 0);
    const float *p3 = base + (ok3 ? ((size_t)h_high * W + w_low) * C : 0);
    const float *p4 = base + (ok4 ? ((size_t)h_high * W + w_high) * C : 0);
    const int cpg = C / groups;
    float *cbase = col + (size_t)t * groups * 9 * cpg;
    for (int c4 = lane; c4 < C4; c4 += 32) {
        const int c = c4 * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside) {                           // (uniform over the half-wave)
            const float4 v1 = *reinterpret_cast<const float4 *>(p1 + c), v2 = *reinterpret_cast<const float4 *>(p2 + c);
            const float4 v3 = *reinterpret_cast<const float4 *>(p3 + c), v4 = *reinterpret_cast<const float4 *>(p4 + c);
            val.x = w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
            val.y = w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
            val.z = w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
            val.w = w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
        const int g = c / cpg, cg = c - g * cpg;
        *reinterpret_cast<float4 *>(cbase + ((size_t)g * 9 + tap) * cpg + cg) = val;
    }
}

// CenterHead second-layer 3x3 convs of all branches in one launch (mmdet3d SeparateHead final conv,
// reached through layers/heads/bev_height_head.py:110).  Output widths are 1..3 channels per branch,
// far below an MFMA tile, and fp32 MFMA runs at the fp32 VALU rate anyway => VALU kernel, register
// blocked: a thread owns 4 vertically adjacent pixels x all (<= 4) outputs of its branch, so one
// ds_read_b128 of the patch feeds up to 3 taps x 4 outputs and every broadcast weight read feeds 4
// pixels (8 FMA per LDS read instead of 2.7).  Workgroup = 128 threads = 32 columns x 16 rows of one
// branch; the (18 x 34) halo patch is staged 16 hidden channels at a time, pixel stride 20 floats
// (80 B: 5*lane mod 16 is a bijection => conflict-free ds_read_b128).
constexpr int kHfTx = 32, kHfTy = 16, kHfRpt = 4, kHfCp = 16, kHfLd = 20;
constexpr int kHfPr = kHfTy + 2, kHfPc = kHfTx + 2;
constexpr int kHfMaxOut = 4;
constexpr int kHfThreads = kHfTx * (kHfTy / kHfRpt);   // 128

__global__ __launch_bounds__(kHfThreads) void head_final_conv_kernel(int H, int W, int nb, int hc, int total_out,
                                                                     const float *__restrict__ hidden,
                                                                     const float *__restrict__ weight,
                                                                     const float *__restrict__ bias,
                                                                     const int32_t *__restrict__ branch_of_out,
                                                                     float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                                   // [18*34][20]
    float *wts = smem + kHfPr * kHfPc * kHfLd;             // [ck][9][hc]
    int *meta = reinterpret_cast<int *>(wts + kHfMaxOut * 9 * hc);
    const int br = blockIdx.z % nb, b = blockIdx.z / nb;
    const int tiles_x = (W + kHfTx - 1) / kHfTx;
    const int ty0 = (blockIdx.x / tiles_x) * kHfTy, tx0 = (blockIdx.x % tiles_x) * kHfTx;
    // which output channels belong to this branch: all threads look in parallel (a serial scan by one
    // thread is ~total_out dependent L2 round trips per workgroup and dominated the first version)
    if (threadIdx.x == 0) { meta[0] = 0x7fffffff; meta[1] = 0; }
    __syncthreads();
    for (int o = threadIdx.x; o < total_out; o += kHfThreads)
        if (branch_of_out[o] == br) { atomicMin(&meta[0], o); atomicAdd(&meta[1], 1); }
    __syncthreads();
    const int first = meta[0];
    const int ck = meta[1] < kHfMaxOut ? meta[1] : kHfMaxOut;
    if (ck == 0) return;
    for (int i = threadIdx.x; i < ck * 9 * hc; i += kHfThreads) wts[i] = weight[(long long)first * 9 * hc + i];
    const int tx = threadIdx.x & (kHfTx - 1), tq = threadIdx.x / kHfTx;   // rows tq*4 .. tq*4+3
    // hidden is [nb][B][H][W][hc]: one NHWC map per branch (pixel stride hc floats, rows contiguous)
    const float *hmap = hidden + ((long long)br * (gridDim.z / nb) + b) * H * W * hc;
    float acc[kHfMaxOut][kHfRpt];
#pragma unroll
    for (int o = 0; o < kHfMaxOut; ++o)
#pragma unroll
        for (int r = 0; r < kHfRpt; ++r) acc[o][r] = 0.f;
    // patch staging is software-pipelined through registers: all of a pass's global loads are issued
    // back to back (one load in flight per thread made the first version latency-bound: 20 dependent
    // HBM round trips per pass) and the NEXT pass is fetched while the current one is multiplied.
    constexpr int kLd = (kHfPr * kHfPc * (kHfCp / 4) + kHfThreads - 1) / kHfThreads;   // float4 per thread per pass
    float4 stage[kLd];
    const float *src[kLd];
#pragma unroll
    for (int it = 0; it < kLd; ++it) {
        const int i = threadIdx.x + it * kHfThreads;
        const int pp = i >> 2, q = i & 3;
        const int yy = ty0 - 1 + pp / kHfPc, xx = tx0 - 1 + pp % kHfPc;
        const bool ok = i < kHfPr * kHfPc * (kHfCp / 4) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        src[it] = ok ? hmap + ((long long)yy * W + xx) * hc + q * 4 : nullptr;
    }
#define SGV3D_HF_LOAD(C0)                                                                              \
    _Pragma("unroll") for (int it = 0; it < kLd; ++it) {                                               \
        float4 t_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
        if (src[it]) t_ = *reinterpret_cast<const float4 *>(src[it] + (C0));                          \
        stage[it].x = t_.x; stage[it].y = t_.y; stage[it].z = t_.z; stage[it].w = t_.w;                \
    }
    SGV3D_HF_LOAD(0);
    for (int c0 = 0; c0 < hc; c0 += kHfCp) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kLd; ++it) {
            const int i = threadIdx.x + it * kHfThreads;
            if (i < kHfPr * kHfPc * (kHfCp / 4))
                *reinterpret_cast<float4 *>(patch + (i >> 2) * kHfLd + (i & 3) * 4) = stage[it];
        }
        __syncthreads();
        if (c0 + kHfCp < hc) { SGV3D_HF_LOAD(c0 + kHfCp); }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int q = 0; q < kHfCp / 4; ++q) {
                float4 p[kHfRpt + 2];
#pragma unroll
                for (int r = 0; r < kHfRpt + 2; ++r)
                    p[r] = *reinterpret_cast<const float4 *>(patch + ((tq * kHfRpt + r) * kHfPc + tx + kx) * kHfLd + q * 4);
#pragma unroll
                for (int o = 0; o < kHfMaxOut; ++o) {
                    if (o < ck) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const float4 wv = *reinterpret_cast<const float4 *>(wts + (o * 9 + ky * 3 + kx) * hc + c0 + q * 4);
#pragma unroll
                            for (int r = 0; r < kHfRpt; ++r)
                                acc[o][r] += p[r + ky].x * wv.x + p[r + ky].y * wv.y + p[r + ky].z * wv.z + p[r + ky].w * wv.w;
                        }
                    }
                }
            }
        }
    }
#undef SGV3D_HF_LOAD
    const int x = tx0 + tx;
    if (x < W) {
#pragma unroll
        for (int o = 0; o < kHfMaxOut; ++o) {
            if (o < ck) {
                const float bv = bias[first + o];
#pragma unroll
                for (int r = 0; r < kHfRpt; ++r) {
                    const int y = ty0 + tq * kHfRpt + r;
                    if (y < H) out[((long long)(b * total_out + first + o) * H + y) * W + x] = acc[o][r] + bv;
                }
            }
        }
    }
}

}  // namespace

extern "C" int sgv3d_maxpool3x3s2(int batch, int in_h, int in_w, int channels, const float *x, float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && channels > 0 && (channels & 3) == 0, "maxpool3x3s2: bad shape");
    SGV3D_REQUIRE(x && y, "maxpool3x3s2: null pointer");
    const int oh = (in_h - 1) / 2 + 1, ow = (in_w - 1) / 2 + 1;
    const long long total = (long long)batch * oh * ow * (channels / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, in_h,
                       in_w, channels / 4, oh, ow, reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y));
    return check_launch("maxpool3x3s2_kernel");
}

extern "C" int sgv3d_maxpool3x3s2_bf16(int batch, int in_h, int in_w, int channels, const void *x, void *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && channels > 0 && channels % 8 == 0, "maxpool3x3s2_bf16: channels must be a multiple of 8");
    SGV3D_REQUIRE(x && y && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                  "maxpool3x3s2_bf16: null or unaligned pointer");
    const int oh = (in_h - 1) / 2 + 1, ow = (in_w - 1) / 2 + 1;
    const long long total = (long long)batch * oh * ow * (channels / 8);
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, in_h,
                       in_w, channels / 8, oh, ow, static_cast<const mp_bf16x8 *>(x), static_cast<mp_bf16x8 *>(y));
    return check_launch("maxpool3x3s2_bf16_kernel");
}

extern "C" int sgv3d_nchw_to_nhwc(int batch, int channels, int h, int w, int c_pad, const float *x, float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && c_pad >= channels, "nchw_to_nhwc: bad shape");
    SGV3D_REQUIRE(x && y, "nchw_to_nhwc: null pointer");
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(batch * hw, kBlock)), dim3(kBlock), 0, as_stream(stream), batch,
                       channels, hw, c_pad, x, y);
    return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int sgv3d_nhwc_to_nchw(int batch, int channels, int h, int w, int ld, int coff, const float *x, float *y,
                                  void *stream) {
    SGV3D_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && ld >= coff + channels && coff >= 0, "nhwc_to_nchw: bad shape");
    SGV3D_REQUIRE(x && y, "nhwc_to_nchw: null pointer");
    const long long hw = (long long)h * w;
    dim3 grid(cdiv(hw, 32), cdiv(channels, 32), batch);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(kBlock), 0, as_stream(stream), channels, hw, ld, coff, x, y);
    return check_launch("nhwc_to_nchw_kernel");
}

extern "C" size_t sgv3d_global_avgpool_workspace_bytes(int batch, int channels) {
    if (batch <= 0 || channels <= 0) return 0;
    return sizeof(float) * (size_t)batch * kAvgChunks * channels;
}

extern "C" int sgv3d_global_avgpool(int batch, int pixels, int channels, int x_ld, const float *x, float *y,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && x_ld >= channels, "global_avgpool: bad shape");
    SGV3D_REQUIRE(x && y && workspace, "global_avgpool: null pointer");
    if (workspace_bytes < sgv3d_global_avgpool_workspace_bytes(batch, channels))
        return fail(SGV3D_ENOSPACE, "global_avgpool: workspace too small");
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(global_avgpool_partial_kernel, dim3(cdiv(channels, 64), kAvgChunks, batch), dim3(kBlock), 0,
                       as_stream(stream), pixels, channels, x_ld, x, part);
    hipLaunchKernelGGL(global_avgpool_final_kernel, dim3(cdiv((long long)batch * channels, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), batch, pixels, channels, part, y);
    return check_launch("global_avgpool_kernel");
}

extern "C" int sgv3d_dense_gated(int batch, int k, int n, const float *x, const float *w, const float *scale,
                                 const float *bias, int act, float *y, const int32_t *run, void *stream) {
    SGV3D_REQUIRE(batch > 0 && k > 0 && n > 0 && act >= 0 && act <= 2, "dense: bad shape/act");
    SGV3D_REQUIRE(x && w && y, "dense: null pointer");
    hipLaunchKernelGGL(dense_kernel, dim3(cdiv((long long)batch * n, kBlock / 64)), dim3(kBlock), 0, as_stream(stream),
                       batch, k, n, x, w, scale, bias, act, y, run);
    return check_launch("dense_kernel");
}

extern "C" int sgv3d_dense(int batch, int k, int n, const float *x, const float *w, const float *scale,
                           const float *bias, int act, float *y, void *stream) {
    return sgv3d_dense_gated(batch, k, n, x, w, scale, bias, act, y, nullptr, stream);
}

extern "C" int sgv3d_broadcast_channels(int batch, int pixels, int channels, int y_ld, int y_coff, const float *v,
                                        float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && y_ld >= y_coff + channels && y_coff >= 0, "broadcast_channels: bad shape");
    SGV3D_REQUIRE(v && y, "broadcast_channels: null pointer");
    const long long total = (long long)batch * pixels * channels;
    hipLaunchKernelGGL(broadcast_channels_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), total,
                       pixels, channels, y_ld, y_coff, v, y);
    return check_launch("broadcast_channels_kernel");
}

extern "C" int sgv3d_scale_channels(int batch, int pixels, int channels, const float *x, const float *gate, float *y,
                                    void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && (channels & 3) == 0, "scale_channels: bad shape");
    SGV3D_REQUIRE(x && gate && y, "scale_channels: null pointer");
    const long long total4 = (long long)batch * pixels * (channels / 4);
    hipLaunchKernelGGL(scale_channels_kernel, dim3(cdiv(total4, kBlock)), dim3(kBlock), 0, as_stream(stream), total4,
                       pixels, channels / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(gate),
                       reinterpret_cast<float4 *>(y));
    return check_launch("scale_channels_kernel");
}

extern "C" int sgv3d_copy_channels(int batch, int pixels, int channels, int x_ld, int x_coff, const float *x, float *y,
                                   void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && channels > 0 && x_coff >= 0 && x_ld >= x_coff + channels, "copy_channels: bad shape");
    SGV3D_REQUIRE(x && y, "copy_channels: null pointer");
    const long long total = (long long)batch * pixels * channels;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), total, channels,
                       x_ld, x_coff, x, y);
    return check_launch("copy_channels_kernel");
}

extern "C" int sgv3d_upsample_bilinear2x(int batch, int h, int w, int channels, const float *x, float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0, "upsample_bilinear2x: bad shape");
    SGV3D_REQUIRE(x && y, "upsample_bilinear2x: null pointer");
    if (channels & 3) {                                  // e.g. the 7-class semantic logits
        const long long n = (long long)batch * 4 * h * w * channels;
        hipLaunchKernelGGL(upsample2x_scalar_kernel, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, h, w,
                           channels, x, y);
        return check_launch("upsample2x_scalar_kernel");
    }
    const long long total = (long long)batch * 4 * h * w * (channels / 4);
    hipLaunchKernelGGL(upsample_bilinear2x_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), batch, h, w,
                       channels / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y));
    return check_launch("upsample_bilinear2x_kernel");
}

extern "C" int sgv3d_add_mul_sigmoid(long long n, const float *a, const float *b, const float *c, float *y, void *stream) {
    SGV3D_REQUIRE(n > 0 && (n & 3) == 0, "add_mul_sigmoid: n must be a positive multiple of 4");
    SGV3D_REQUIRE(a && b && c && y, "add_mul_sigmoid: null pointer");
    hipLaunchKernelGGL(add_mul_sigmoid_kernel, dim3(cdiv(n / 4, kBlock)), dim3(kBlock), 0, as_stream(stream), n / 4,
                       reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b),
                       reinterpret_cast<const float4 *>(c), reinterpret_cast<float4 *>(y));
    return check_launch("add_mul_sigmoid_kernel");
}

extern "C" int sgv3d_bsm_compose(int batch, int pixels, int num_depth, int context_channels, int semantic_channels,
                                 int ld, const float *semantic_logits, int semantic_ld, float background_threshold,
                                 float *height_context, void *stream) {
    SGV3D_REQUIRE(batch > 0 && pixels > 0 && num_depth > 0 && context_channels > 0 && semantic_channels > 0 &&
                      ld >= num_depth + context_channels + semantic_channels && semantic_ld >= semantic_channels,
                  "bsm_compose: bad shape");
    SGV3D_REQUIRE(semantic_logits && height_context, "bsm_compose: null pointer");
    const long long px = (long long)batch * pixels;
    hipLaunchKernelGGL(bsm_compose_kernel, dim3(cdiv(px, kBlock / 32)), dim3(kBlock), 0, as_stream(stream), px, num_depth,
                       context_channels, semantic_channels, ld, semantic_logits, semantic_ld, background_threshold,
                       height_context);
    return check_launch("bsm_compose_kernel");
}

extern "C" int sgv3d_deform_im2col3x3(int batch, int h, int w, int channels, int groups, const float *x,
                                      const float *offset, int off_ld, float *col, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0 && groups > 0 && channels % groups == 0 &&
                      ((channels / groups) & 3) == 0 && off_ld >= 18,
                  "deform_im2col3x3: bad shape");
    SGV3D_REQUIRE(x && offset && col, "deform_im2col3x3: null pointer");
    const long long pairs = (long long)batch * h * w * 9;           // one half-wave per (pixel, tap)
    SGV3D_REQUIRE(pairs < 0x7fffffffLL, "deform_im2col3x3: too many pixels");
    hipLaunchKernelGGL(deform_im2col3x3_kernel, dim3(cdiv(pairs, kBlock / 32)), dim3(kBlock), 0, as_stream(stream), batch, h, w,
                       channels, groups, x, offset, off_ld, col);
    return check_launch("deform_im2col3x3_kernel");
}

extern "C" int sgv3d_head_final_conv(int batch, int h, int w, int num_branches, int hidden_ch, int total_out,
                                     const float *hidden, const float *weight, const float *bias,
                                     const int32_t *branch_of_out, float *out, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && num_branches > 0 && hidden_ch > 0 && hidden_ch % kHfCp == 0 && total_out > 0,
                  "head_final_conv: bad shape");
    SGV3D_REQUIRE(hidden && weight && bias && branch_of_out && out, "head_final_conv: null pointer");
    SGV3D_REQUIRE((long long)batch * num_branches <= 65535, "head_final_conv: batch*branches exceeds grid.z");
    const size_t lds = sizeof(float) * ((size_t)kHfPr * kHfPc * kHfLd + (size_t)kHfMaxOut * 9 * hidden_ch + 4);
    static PerDeviceSize attr_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&head_final_conv_kernel), 96 * 1024, attr_set))
        return fail(SGV3D_ELAUNCH, "head_final_conv: cannot raise the dynamic LDS limit");
    SGV3D_REQUIRE(lds <= 96 * 1024, "head_final_conv: hidden_ch too large");
    const int tiles = cdiv(h, kHfTy) * cdiv(w, kHfTx);
    dim3 grid(tiles, 1, batch * num_branches);
    hipLaunchKernelGGL(head_final_conv_kernel, grid, dim3(kHfThreads), lds, as_stream(stream), h, w, num_branches, hidden_ch,
                       total_out, hidden, weight, bias, branch_of_out, out);
    return check_launch("head_final_conv_kernel");
}
