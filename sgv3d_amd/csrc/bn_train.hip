// Training-mode BatchNorm2d fused with the residual add and the ReLU around it, NHWC fp32.
// Reference: every `norm(conv(x))` / `relu(out + identity)` of the mmdet ResNet blocks, SECONDFPN, HeightNet and the
// head (layers/backbones/lss_fpn.py, layers/heads/bev_height_head.py) in training mode, i.e. ATen / MIOpen batch_norm
// with batch statistics + separate add and ReLU kernels.  All four kernels are HBM-bound streams:
//
//   forward   bn_stats_kernel    per-channel sum and sum of squares over the B*H*W pixels (float64 accumulators: no
//                                cancellation in E[x^2] - mean^2), per-block partials
//             bn_finalize_kernel mean, biased variance, 1/sqrt(var + eps), folded scale / shift, running statistics
//                                (momentum, unbiased variance) -- what nn.BatchNorm2d updates in training mode
//             bn_apply_kernel    y = relu(x * scale + shift + residual)              (reads x [+ residual], writes y)
//   backward  bn_bwd_reduce_kernel   dz = dy * (y > 0);  dbeta = sum dz;  dgamma = sum dz * xhat   (float64 partials)
//             bn_bwd_finalize_kernel sums the partials
//             bn_bwd_apply_kernel    dx = gamma * invstd * (dz - dbeta / M - xhat * dgamma / M);  d_residual = dz
//
// Thread layout: 16 lanes x float4 cover 64 channels of a pixel (256 contiguous bytes), 16 pixel rows per block pass;
// a block owns a (64-channel, pixel-range) slab, blockIdx.x = channel group, blockIdx.y = pixel range.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kMaxBlocks = 1024;  // partial blocks (channel groups x pixel ranges) the workspace is sized for

struct BnArgs {
    const float *x, *res, *y_in, *dy;     // forward: x, res;  backward: x, y_in (forward output, ReLU mask), dy
    float *y, *dx, *dres;
    const float *gamma, *beta;
    float *running_mean, *running_var, *mean, *invstd, *scale, *shift, *dgamma, *dbeta;
    double *partial;                       // [ranges][C][2]
    long long pixels;
    int channels, ranges, pix_per_range, relu;   // relu 2 (backward): the mask is bn(x) > 0 recomputed from x (no residual in the forward)
    float momentum, eps;
};

__global__ __launch_bounds__(256) void bn_stats_kernel(const BnArgs a) {
    __shared__ double lds[16][64][2];
    const int cg = blockIdx.x * 64, c4 = (threadIdx.x & 15) * 4, prow = threadIdx.x >> 4;
    const long long p0 = (long long)blockIdx.y * a.pix_per_range;
    const long long p1 = p0 + a.pix_per_range < a.pixels ? p0 + a.pix_per_range : a.pixels;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (cg + c4 < a.channels) {
        const float *base = a.x + cg + c4;
        long long p = p0 + prow;
        for (; p + 48 < p1; p += 64) {           // four loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(base + (p + 16 * u) * a.channels);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
                q[0] += (double)v[u].x * v[u].x; q[1] += (double)v[u].y * v[u].y;
                q[2] += (double)v[u].z * v[u].z; q[3] += (double)v[u].w * v[u].w;
            }
        }
        for (; p < p1; p += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(base + p * a.channels);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { lds[prow][c4 + i][0] = s[i]; lds[prow][c4 + i][1] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0;
        for (int r = 0; r < 16; ++r) t += lds[r][c][k];
        if (cg + c < a.channels) a.partial[((size_t)blockIdx.y * a.channels + cg + c) * 2 + k] = t;
    }
}

// Sum of the per-range partials of channel c: kFinCh channels per block, 256 / kFinCh threads per channel over the range axis (a
// 1024-range layer: 16 independent loads per thread, issued together), the slice sums added in slice order (fixed order);
// s0 / s1 valid in the threads with slice 0.
constexpr int kFinCh = 4, kFinSlices = 256 / kFinCh;

__device__ __forceinline__ void sum_partials(const BnArgs &a, int c, int slice, double (*lds)[kFinCh][2], double &s0, double &s1) {
    double s = 0, q = 0;
    if (c < a.channels) {
        int r = slice;
        for (; r + 3 * kFinSlices < a.ranges; r += 4 * kFinSlices) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const double2 *>(a.partial + ((size_t)(r + u * kFinSlices) * a.channels + c) * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += v[u].x; q += v[u].y; }
        }
        for (; r < a.ranges; r += kFinSlices) {
            const double2 v = *reinterpret_cast<const double2 *>(a.partial + ((size_t)r * a.channels + c) * 2);
            s += v.x;
            q += v.y;
        }
    }
    const int cl = threadIdx.x % kFinCh;
    lds[slice][cl][0] = s;
    lds[slice][cl][1] = q;
    __syncthreads();
    s0 = s1 = 0;
    if (slice == 0)
        for (int i = 0; i < kFinSlices; ++i) { s0 += lds[i][cl][0]; s1 += lds[i][cl][1]; }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnArgs a) {
    __shared__ double lds[kFinSlices][kFinCh][2];
    const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, slice = threadIdx.x / kFinCh;
    double s, q;
    sum_partials(a, c, slice, lds, s, q);
    if (slice != 0 || c >= a.channels) return;
    const double m = (double)a.pixels;
    const double mean = s / m;
    double var = q / m - mean * mean;
    var = var > 0 ? var : 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    a.mean[c] = (float)mean;
    a.invstd[c] = invstd;
    const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
    // (explicit roundings: the backward recomputes exactly these two numbers when it derives the ReLU mask from x)
    const float scale = __fmul_rn(g, invstd);
    a.scale[c] = scale;
    a.shift[c] = __fmaf_rn(-(float)mean, scale, b);
    if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
    if (a.running_var) {
        const double unbiased = m > 1 ? var * m / (m - 1) : var;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const BnArgs a) {
    const int c4n = a.channels / 4;
    const long long total = a.pixels * c4n;
    const float4 *x4 = reinterpret_cast<const float4 *>(a.x), *r4 = reinterpret_cast<const float4 *>(a.res);
    const float4 *sc4 = reinterpret_cast<const float4 *>(a.scale), *sh4 = reinterpret_cast<const float4 *>(a.shift);
    float4 *y4 = reinterpret_cast<float4 *>(a.y);
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    // the grid is sized so that its thread count is a multiple of channels / 4: a thread keeps its channels
    const int c = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)c4n);
    const float4 sc = sc4[c], sh = sh4[c];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        const float4 v = x4[i];
        float4 o = make_float4(__fmaf_rn(v.x, sc.x, sh.x), __fmaf_rn(v.y, sc.y, sh.y), __fmaf_rn(v.z, sc.z, sh.z), __fmaf_rn(v.w, sc.w, sh.w));
        if (r4) {
            const float4 r = r4[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        o.x = fmaxf(o.x, floor_); o.y = fmaxf(o.y, floor_); o.z = fmaxf(o.z, floor_); o.w = fmaxf(o.w, floor_);
        y4[i] = o;
    }
}

// scale / shift of channels c .. c + 3 exactly as bn_finalize_kernel folded them (same roundings)
__device__ __forceinline__ void fold_affine(const BnArgs &a, int c, const float4 mean, const float4 istd, float4 &sc, float4 &sh) {
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.gamma) g = *reinterpret_cast<const float4 *>(a.gamma + c);
    if (a.beta) b = *reinterpret_cast<const float4 *>(a.beta + c);
    sc = make_float4(__fmul_rn(g.x, istd.x), __fmul_rn(g.y, istd.y), __fmul_rn(g.z, istd.z), __fmul_rn(g.w, istd.w));
    sh = make_float4(__fmaf_rn(-mean.x, sc.x, b.x), __fmaf_rn(-mean.y, sc.y, b.y), __fmaf_rn(-mean.z, sc.z, b.z), __fmaf_rn(-mean.w, sc.w, b.w));
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnArgs a) {
    __shared__ double lds[16][64][2];
    const int cg = blockIdx.x * 64, c4 = (threadIdx.x & 15) * 4, prow = threadIdx.x >> 4;
    const long long p0 = (long long)blockIdx.y * a.pix_per_range;
    const long long p1 = p0 + a.pix_per_range < a.pixels ? p0 + a.pix_per_range : a.pixels;
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    if (cg + c4 < a.channels) {
        const float4 mean = *reinterpret_cast<const float4 *>(a.mean + cg + c4);
        const float4 istd = *reinterpret_cast<const float4 *>(a.invstd + cg + c4);
        float4 fsc, fsh;
        const bool from_x = a.relu == 2;
        if (from_x) fold_affine(a, cg + c4, mean, istd, fsc, fsh);
        auto one = [&](float4 d, const float4 v, float4 y) {
            if (from_x) y = make_float4(__fmaf_rn(v.x, fsc.x, fsh.x), __fmaf_rn(v.y, fsc.y, fsh.y), __fmaf_rn(v.z, fsc.z, fsh.z), __fmaf_rn(v.w, fsc.w, fsh.w));
            if (a.relu) {
                d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f; d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
            }
            sb[0] += d.x; sb[1] += d.y; sb[2] += d.z; sb[3] += d.w;
            sg[0] += (double)d.x * ((v.x - mean.x) * istd.x); sg[1] += (double)d.y * ((v.y - mean.y) * istd.y);
            sg[2] += (double)d.z * ((v.z - mean.z) * istd.z); sg[3] += (double)d.w * ((v.w - mean.w) * istd.w);
        };
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        long long p = p0 + prow;
        for (; p + 16 < p1; p += 32) {           // six loads in flight per thread
            const size_t o0 = p * a.channels + cg + c4, o1 = (p + 16) * a.channels + cg + c4;
            const float4 d0 = *reinterpret_cast<const float4 *>(a.dy + o0), d1 = *reinterpret_cast<const float4 *>(a.dy + o1);
            const float4 v0 = *reinterpret_cast<const float4 *>(a.x + o0), v1 = *reinterpret_cast<const float4 *>(a.x + o1);
            const float4 y0 = a.relu == 1 ? *reinterpret_cast<const float4 *>(a.y_in + o0) : zero;
            const float4 y1 = a.relu == 1 ? *reinterpret_cast<const float4 *>(a.y_in + o1) : zero;
            one(d0, v0, y0);
            one(d1, v1, y1);
        }
        for (; p < p1; p += 16) {
            const size_t o = p * a.channels + cg + c4;
            one(*reinterpret_cast<const float4 *>(a.dy + o), *reinterpret_cast<const float4 *>(a.x + o),
                a.relu == 1 ? *reinterpret_cast<const float4 *>(a.y_in + o) : zero);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { lds[prow][c4 + i][0] = sb[i]; lds[prow][c4 + i][1] = sg[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0;
        for (int r = 0; r < 16; ++r) t += lds[r][c][k];
        if (cg + c < a.channels) a.partial[((size_t)blockIdx.y * a.channels + cg + c) * 2 + k] = t;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BnArgs a) {
    __shared__ double lds[kFinSlices][kFinCh][2];
    const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, slice = threadIdx.x / kFinCh;
    double sb, sg;
    sum_partials(a, c, slice, lds, sb, sg);
    if (slice != 0 || c >= a.channels) return;
    a.dbeta[c] = (float)sb;
    a.dgamma[c] = (float)sg;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnArgs a) {
    const int c4n = a.channels / 4;
    const long long total = a.pixels * c4n;
    const float inv_m = (float)(1.0 / (double)a.pixels);
    const float4 *x4 = reinterpret_cast<const float4 *>(a.x), *y4 = reinterpret_cast<const float4 *>(a.y_in);
    const float4 *d4 = reinterpret_cast<const float4 *>(a.dy);
    float4 *dx4 = reinterpret_cast<float4 *>(a.dx), *dr4 = reinterpret_cast<float4 *>(a.dres);
    const int c = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)c4n) * 4;   // constant per thread (grid sizing)
    const float4 mean = *reinterpret_cast<const float4 *>(a.mean + c), istd = *reinterpret_cast<const float4 *>(a.invstd + c);
    float4 db = *reinterpret_cast<const float4 *>(a.dbeta + c), dg = *reinterpret_cast<const float4 *>(a.dgamma + c);
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.gamma) g = *reinterpret_cast<const float4 *>(a.gamma + c);
    // dx = k1 * (dz - kb - (x - mean) * kg) with k1 = gamma * invstd, kb = dbeta / M, kg = invstd * dgamma / M
    const float4 k1 = make_float4(g.x * istd.x, g.y * istd.y, g.z * istd.z, g.w * istd.w);
    db.x *= inv_m; db.y *= inv_m; db.z *= inv_m; db.w *= inv_m;
    dg.x *= istd.x * inv_m; dg.y *= istd.y * inv_m; dg.z *= istd.z * inv_m; dg.w *= istd.w * inv_m;
    float4 fsc, fsh;
    const bool from_x = a.relu == 2;
    if (from_x) fold_affine(a, c, mean, istd, fsc, fsh);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        float4 d = d4[i];
        const float4 v = x4[i];
        if (a.relu) {
            const float4 y = from_x ? make_float4(__fmaf_rn(v.x, fsc.x, fsh.x), __fmaf_rn(v.y, fsc.y, fsh.y), __fmaf_rn(v.z, fsc.z, fsh.z),
                                                  __fmaf_rn(v.w, fsc.w, fsh.w))
                                    : y4[i];
            d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f; d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
        }
        if (dr4) dr4[i] = d;
        float4 o;
        o.x = k1.x * (d.x - db.x - (v.x - mean.x) * dg.x);
        o.y = k1.y * (d.y - db.y - (v.y - mean.y) * dg.y);
        o.z = k1.z * (d.z - db.z - (v.z - mean.z) * dg.z);
        o.w = k1.w * (d.w - db.w - (v.w - mean.w) * dg.w);
        dx4[i] = o;
    }
}

int plan(long long pixels, int channels, BnArgs &a) {
    SGV3D_REQUIRE(pixels > 0 && channels > 0 && channels % 4 == 0, "batchnorm: pixels > 0 and channels %% 4 == 0 required");
    a.pixels = pixels; a.channels = channels;
    SGV3D_REQUIRE(cdiv(channels, 64) <= kMaxBlocks, "batchnorm: too many channels");
    // ~1024 blocks of 4 waves with 4 - 6 loads in flight per thread, at least 128 pixels per range
    int ranges = kMaxBlocks / cdiv(channels, 64);
    ranges = ranges < 1 ? 1 : ranges;
    const long long cap = pixels / 128 > 0 ? pixels / 128 : 1;
    ranges = ranges > cap ? (int)cap : ranges;
    a.pix_per_range = (int)((pixels + ranges - 1) / ranges);
    a.ranges = (int)((pixels + a.pix_per_range - 1) / a.pix_per_range);
    return SGV3D_OK;
}

// Blocks of the elementwise passes: about `total / 256` capped at 8192, rounded up to a multiple of c4n / gcd(c4n, 256) so
// that the thread count is a multiple of c4n (= channels / 4) and every thread keeps its channel chunk.
int stream_blocks(long long total, int c4n) {
    long long b = (total + 255) / 256;
    b = b < 8192 ? b : 8192;
    int g = c4n, h = 256;
    while (h) { const int t = g % h; g = h; h = t; }
    const int step = c4n / g;
    return (int)((b + step - 1) / step * step);
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t partial_bytes(int) { return (size_t)kMaxBlocks * 64 * 2 * sizeof(double); }

}  // namespace

extern "C" size_t sgv3d_batchnorm_workspace_bytes(int channels) {
    return channels > 0 ? partial_bytes(channels) + 2 * (size_t)channels * sizeof(float) : 0;
}

extern "C" int sgv3d_batchnorm_train_forward(long long pixels, int channels, const float *x, const float *residual,
                                             const float *gamma, const float *beta, float *running_mean,
                                             float *running_var, float momentum, float eps, int relu, float *y,
                                             float *save_mean, float *save_invstd, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    BnArgs a{};
    if (int rc = plan(pixels, channels, a)) return rc;
    SGV3D_REQUIRE(x && y && save_mean && save_invstd && workspace, "batchnorm_train_forward: null pointer");
    SGV3D_REQUIRE(workspace_bytes >= sgv3d_batchnorm_workspace_bytes(channels), "batchnorm_train_forward: workspace too small");
    SGV3D_REQUIRE(aligned16(x) && aligned16(y) && (!residual || aligned16(residual)) && aligned16(workspace),
                  "batchnorm_train_forward: buffers must be 16-byte aligned");
    a.x = x; a.res = residual; a.y = y; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean;
    a.running_var = running_var; a.mean = save_mean; a.invstd = save_invstd; a.momentum = momentum; a.eps = eps; a.relu = relu;
    a.partial = static_cast<double *>(workspace);
    a.scale = reinterpret_cast<float *>(static_cast<char *>(workspace) + partial_bytes(channels));
    a.shift = a.scale + channels;
    hipStream_t s = as_stream(stream);
    bn_stats_kernel<<<dim3(cdiv(channels, 64), a.ranges), 256, 0, s>>>(a);
    if (int rc = check_launch("bn_stats_kernel")) return rc;
    bn_finalize_kernel<<<cdiv(channels, kFinCh), 256, 0, s>>>(a);
    if (int rc = check_launch("bn_finalize_kernel")) return rc;
    bn_apply_kernel<<<stream_blocks(pixels * (channels / 4), channels / 4), 256, 0, s>>>(a);
    return check_launch("bn_apply_kernel");
}

namespace {
int bn_backward(long long pixels, int channels, const float *x, const float *y, const float *dy, const float *gamma, const float *beta,
                const float *save_mean, const float *save_invstd, int relu, float *dx, float *dresidual, float *dgamma, float *dbeta,
                void *workspace, size_t workspace_bytes, void *stream);
}

extern "C" int sgv3d_batchnorm_train_backward(long long pixels, int channels, const float *x, const float *y,
                                              const float *dy, const float *gamma, const float *save_mean,
                                              const float *save_invstd, int relu, float *dx, float *dresidual,
                                              float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                              void *stream) {
    SGV3D_REQUIRE(!relu || y, "batchnorm_train_backward: the forward output is needed for the ReLU mask");
    return bn_backward(pixels, channels, x, y, dy, gamma, nullptr, save_mean, save_invstd, relu ? 1 : 0, dx, dresidual, dgamma, dbeta,
                       workspace, workspace_bytes, stream);
}

extern "C" int sgv3d_batchnorm_relu_train_backward_from_x(long long pixels, int channels, const float *x, const float *dy,
                                                          const float *gamma, const float *beta, const float *save_mean,
                                                          const float *save_invstd, float *dx, float *dgamma, float *dbeta,
                                                          void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE((!gamma || aligned16(gamma)) && (!beta || aligned16(beta)), "batchnorm_relu_train_backward_from_x: gamma / beta must be 16-byte aligned");
    return bn_backward(pixels, channels, x, nullptr, dy, gamma, beta, save_mean, save_invstd, 2, dx, nullptr, dgamma, dbeta, workspace,
                       workspace_bytes, stream);
}

namespace {
int bn_backward(long long pixels, int channels, const float *x, const float *y, const float *dy, const float *gamma, const float *beta,
                const float *save_mean, const float *save_invstd, int relu, float *dx, float *dresidual, float *dgamma, float *dbeta,
                void *workspace, size_t workspace_bytes, void *stream) {
    BnArgs a{};
    if (int rc = plan(pixels, channels, a)) return rc;
    SGV3D_REQUIRE(x && dy && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "batchnorm_train_backward: null pointer");
    a.beta = beta;
    SGV3D_REQUIRE(workspace_bytes >= sgv3d_batchnorm_workspace_bytes(channels), "batchnorm_train_backward: workspace too small");
    SGV3D_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && (!y || aligned16(y)) && (!dresidual || aligned16(dresidual)) &&
                  aligned16(save_mean) && aligned16(save_invstd) && aligned16(dgamma) && aligned16(dbeta) && (!gamma || aligned16(gamma)),
                  "batchnorm_train_backward: buffers must be 16-byte aligned");
    a.x = x; a.y_in = y; a.dy = dy; a.gamma = gamma; a.mean = const_cast<float *>(save_mean);
    a.invstd = const_cast<float *>(save_invstd); a.relu = relu; a.dx = dx; a.dres = dresidual; a.dgamma = dgamma; a.dbeta = dbeta;
    a.partial = static_cast<double *>(workspace);
    hipStream_t s = as_stream(stream);
    bn_bwd_reduce_kernel<<<dim3(cdiv(channels, 64), a.ranges), 256, 0, s>>>(a);
    if (int rc = check_launch("bn_bwd_reduce_kernel")) return rc;
    bn_bwd_finalize_kernel<<<cdiv(channels, kFinCh), 256, 0, s>>>(a);
    if (int rc = check_launch("bn_bwd_finalize_kernel")) return rc;
    bn_bwd_apply_kernel<<<stream_blocks(pixels * (channels / 4), channels / 4), 256, 0, s>>>(a);
    return check_launch("bn_bwd_apply_kernel");
}
}  // namespace
