// Training side of the SGV3D BSM branch: the semantic supervision of
// exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:258-305 and the gradients of the two BSM-only
// elementwise layers.
//
//   sgv3d_semantic_labels_downsample   get_downsampled_gt_semantic (:258-276): the class id of a feature cell is the
//                                      maximum id over its factor x factor block of the mask image
//   sgv3d_focal_loss_with_logits       losses/focal.py:57-90 + losses/_functional.py:37-108 (alpha, gamma; modes
//                                      binary / multilabel / multiclass, ignore_index, reduction mean | sum): value and
//                                      d loss / d logit in ONE pass over the logits; block partial sums in float64,
//                                      added in a fixed order (bitwise repeatable, no float atomics)
//   sgv3d_upsample_bilinear2x_backward adjoint of F.interpolate(scale_factor=2, 'bilinear') (TaskFPN,
//                                      layers/backbones/bsm_lss_fpn.py:210, and get_loss :293) in gather form
//   sgv3d_add_mul_sigmoid_backward     gradients of y = a + b * sigmoid(c) (SABlock + residual, bsm_lss_fpn.py:159,211)
//
// All of it is HBM-bound streaming work (one read of the operands, one write of the results).
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kT = 256;
constexpr int kFocalBlocks = 512;

__device__ __forceinline__ double block_sum_f64(double v, double *lds) {   // kT threads; result valid in thread 0
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kT / 64; ++i) r += lds[i];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------- labels
__global__ void __launch_bounds__(kT) labels_downsample_kernel(int B, int H, int W, int f, const unsigned char *__restrict__ gt,
                                                               unsigned char *__restrict__ out) {
    const int oh = H / f, ow = W / f;
    const long long total = (long long)B * oh * ow;
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % ow);
    const long long t = i / ow;
    const int y = (int)(t % oh);
    const int b = (int)(t / oh);
    const unsigned char *p = gt + ((long long)b * H + (long long)y * f) * W + (long long)x * f;
    unsigned int m = 0;
    if (f == 8 && (W & 7) == 0) {                       // the shipped configs: one 8-byte load per row of the block
        for (int r = 0; r < 8; ++r) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p + (long long)r * W);
            unsigned int w0 = v.x, w1 = v.y;
            for (int k = 0; k < 4; ++k) {
                m = max(m, (w0 >> (8 * k)) & 255u);
                m = max(m, (w1 >> (8 * k)) & 255u);
            }
        }
    } else {
        for (int r = 0; r < f; ++r)
            for (int c = 0; c < f; ++c) m = max(m, (unsigned int)p[(long long)r * W + c]);
    }
    out[i] = (unsigned char)m;
}

// ---------------------------------------------------------------------------------------------- focal loss
struct FocalArgs {
    const float *logit;          // element (b, c, p) at b*sb + c*sc + p*sp
    float *grad;                 // same addressing (or null)
    const void *target;          // label_kind 0: uint8 [B][P]; 1: int64 [B][P]; 2: float32 addressed like logit
    long long sb, sc, sp;
    int batch, classes, pixels, label_kind;
    int use_alpha, use_ignore, mean;
    long long ignore_index;
    float alpha, gamma, grad_scale;
    double *partial;             // [kFocalBlocks] loss sums | [kFocalBlocks] kept-pixel counts
    float *out;                  // [1]
};

__device__ __forceinline__ float focal_elem(float x, float t, const FocalArgs &a, float &dldx) {
    // F.binary_cross_entropy_with_logits: max(x, 0) - x*t + log1p(exp(-|x|))   (_functional.py:69)
    const float e = expf(-fabsf(x));
    const float logpt = fmaxf(x, 0.f) - x * t + log1pf(e);
    const float pt = expf(-logpt);                                              // :70
    const float om = 1.f - pt;
    float focal, dfocal;                                                        // (1 - pt)^gamma and its derivative in (1 - pt)
    if (a.gamma == 2.f) { focal = om * om; dfocal = 2.f * om; }
    else if (a.gamma == 1.f) { focal = om; dfocal = 1.f; }
    else if (a.gamma == 0.f) { focal = 1.f; dfocal = 0.f; }
    else { focal = powf(om, a.gamma); dfocal = om > 0.f ? a.gamma * powf(om, a.gamma - 1.f) : 0.f; }
    const float w = a.use_alpha ? a.alpha * t + (1.f - a.alpha) * (1.f - t) : 1.f;   // :82-83
    // d logpt / dx = sigmoid(x) - t;  d(1 - pt)/dx = pt * (sigmoid(x) - t)
    const float s = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dldx = w * (s - t) * (focal + logpt * dfocal * pt);
    return focal * logpt * w;
}

__global__ void __launch_bounds__(kT) focal_count_kernel(FocalArgs a) {
    __shared__ double lds[kT / 64];
    double n = 0;
    const long long total = (long long)a.batch * a.pixels;
    for (long long e = blockIdx.x * (long long)kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
        long long lab;
        if (a.label_kind == 0) lab = static_cast<const unsigned char *>(a.target)[e];
        else lab = static_cast<const long long *>(a.target)[e];
        n += (a.use_ignore && lab == a.ignore_index) ? 0.0 : 1.0;
    }
    const double s = block_sum_f64(n, lds);
    if (threadIdx.x == 0) a.partial[kFocalBlocks + blockIdx.x] = s;
}

// One thread per (sample, pixel); the classes of the pixel are walked in the thread (7 for the BSM maps), so a label is
// read once.  Channel-last maps (sc == 1) put a pixel's classes in one 28-byte run; plane maps (sp == 1) are coalesced
// across the threads per class.
__global__ void __launch_bounds__(kT) focal_kernel(FocalArgs a) {
    __shared__ double lds[kT / 64];
    // the divisor of reduction='mean' is known before the gradients are written (fixed-order sum of the counts)
    __shared__ double s_div;
    if (a.mean) {
        double c = 0;
        if (a.label_kind == 2) {
            c = threadIdx.x == 0 ? (double)a.batch * a.pixels * a.classes : 0.0;
        } else {
            for (int i = threadIdx.x; i < kFocalBlocks; i += kT) c += a.partial[kFocalBlocks + i];
        }
        const double tot = block_sum_f64(c, lds);
        if (threadIdx.x == 0) s_div = tot;
        __syncthreads();
    }
    const float gs = a.mean ? (float)((double)a.grad_scale / fmax(s_div, 0.0)) : a.grad_scale;   // 0 kept pixels: nan, as torch's mean of nothing
    double acc = 0;
    const long long total = (long long)a.batch * a.pixels;
    for (long long e = blockIdx.x * (long long)kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
        const long long b = e / a.pixels, p = e - b * a.pixels;
        const long long base = b * a.sb + p * a.sp;
        long long lab = 0;
        bool keep = true;
        if (a.label_kind == 0) lab = static_cast<const unsigned char *>(a.target)[e];
        else if (a.label_kind == 1) lab = static_cast<const long long *>(a.target)[e];
        if (a.label_kind != 2 && a.use_ignore && lab == a.ignore_index) keep = false;
        for (int c = 0; c < a.classes; ++c) {
            const long long off = base + c * a.sc;
            float g = 0.f;
            if (keep) {
                const float t = a.label_kind == 2 ? static_cast<const float *>(a.target)[off] : (lab == c ? 1.f : 0.f);
                acc += (double)focal_elem(a.logit[off], t, a, g);
            }
            if (a.grad) a.grad[off] = g * gs;
        }
    }
    const double s = block_sum_f64(acc, lds);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kT) focal_finish_kernel(FocalArgs a) {
    __shared__ double lds[kT / 64];
    double v = 0, c = 0;
    for (int i = threadIdx.x; i < kFocalBlocks; i += kT) {
        v += a.partial[i];
        if (a.label_kind != 2) c += a.partial[kFocalBlocks + i];
    }
    const double vs = block_sum_f64(v, lds);
    const double cs = block_sum_f64(c, lds);
    if (threadIdx.x == 0) {
        const double div = a.label_kind == 2 ? (double)a.batch * a.pixels * a.classes : cs;
        a.out[0] = (float)(a.mean ? vs / div : vs);
    }
}

// ---------------------------------------------------------------------------------------------- upsample backward
// Forward (misc_layers.hip: upsample_bilinear2x_kernel): output row r reads source rows floor(s), floor(s)+1 (clamped to
// H-1) with s = max(r/2 - 1/4, 0).  Per axis that is: r = 2i -> 1/4 x[i-1] + 3/4 x[i] (r = 0: x[0] alone),
// r = 2i+1 -> 3/4 x[i] + 1/4 x[i+1] (last row: x[H-1] alone).  The adjoint, gathered per source row i:
//   dx[i] = 3/4 (dy[2i] + dy[2i+1]) + 1/4 dy[2i-1] (i > 0) + 1/4 dy[2i+2] (i < H-1)
//           + 1/4 dy[0] (i == 0) + 1/4 dy[2H-1] (i == H-1)
__device__ __forceinline__ int up_taps(int i, int n, int idx[4], float wt[4]) {
    int k = 0;
    idx[k] = 2 * i; wt[k++] = (i == 0) ? 1.f : 0.75f;
    idx[k] = 2 * i + 1; wt[k++] = (i == n - 1) ? 1.f : 0.75f;
    if (i > 0) { idx[k] = 2 * i - 1; wt[k++] = 0.25f; }
    if (i < n - 1) { idx[k] = 2 * i + 2; wt[k++] = 0.25f; }
    return k;
}

__global__ void __launch_bounds__(kT) upsample2x_backward_kernel(int B, int H, int W, int C, const float *__restrict__ dy,
                                                                 float *__restrict__ dx) {
    const long long total = (long long)B * H * W * C;
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    int hi[4], wi[4];
    float hw[4], ww[4];
    const int nh = up_taps(h, H, hi, hw), nw = up_taps(w, W, wi, ww);
    const float *p = dy + (long long)b * 4 * H * W * C + c;
    float acc = 0.f;
    for (int r = 0; r < nh; ++r) {
        float row = 0.f;
        for (int q = 0; q < nw; ++q) row += ww[q] * p[((long long)hi[r] * 2 * W + wi[q]) * C];
        acc += hw[r] * row;
    }
    dx[i] = acc;
}

// ---------------------------------------------------------------------------------------------- SABlock backward
__global__ void __launch_bounds__(kT) add_mul_sigmoid_backward_kernel(long long n4, const float4 *__restrict__ dy,
                                                                      const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                                      float4 *__restrict__ db, float4 *__restrict__ dc) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= n4) return;
    const float4 g = dy[i], bv = b[i], cv = c[i];
    float4 ob, oc;
    float s;
    s = 1.f / (1.f + expf(-cv.x)); ob.x = g.x * s; oc.x = g.x * bv.x * s * (1.f - s);
    s = 1.f / (1.f + expf(-cv.y)); ob.y = g.y * s; oc.y = g.y * bv.y * s * (1.f - s);
    s = 1.f / (1.f + expf(-cv.z)); ob.z = g.z * s; oc.z = g.z * bv.z * s * (1.f - s);
    s = 1.f / (1.f + expf(-cv.w)); ob.w = g.w * s; oc.w = g.w * bv.w * s * (1.f - s);
    db[i] = ob;
    dc[i] = oc;
}

}  // namespace

extern "C" int sgv3d_semantic_labels_downsample(int batch, int h, int w, int factor, const unsigned char *gt,
                                                unsigned char *labels, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && factor > 0 && h % factor == 0 && w % factor == 0,
                  "semantic_labels_downsample: the image must be a whole number of factor x factor blocks");
    SGV3D_REQUIRE(gt && labels, "semantic_labels_downsample: null pointer");
    const long long total = (long long)batch * (h / factor) * (w / factor);
    labels_downsample_kernel<<<cdiv(total, kT), kT, 0, as_stream(stream)>>>(batch, h, w, factor, gt, labels);
    return check_launch("labels_downsample_kernel");
}

extern "C" size_t sgv3d_focal_loss_workspace_bytes(void) { return (size_t)kFocalBlocks * 2 * sizeof(double); }

extern "C" int sgv3d_focal_loss_with_logits(int batch, int num_classes, int pixels, const float *logits,
                                            long long batch_stride, long long class_stride, long long pixel_stride,
                                            const void *target, int target_kind, float alpha, float gamma,
                                            long long ignore_index, int use_ignore_index, int reduction_mean,
                                            float grad_scale, float *grad, float *loss_out, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && num_classes > 0 && pixels > 0, "focal_loss: bad sizes");
    SGV3D_REQUIRE(logits && target && loss_out && workspace, "focal_loss: null pointer");
    SGV3D_REQUIRE(target_kind >= 0 && target_kind <= 2, "focal_loss: target_kind must be 0 (uint8 labels), 1 (int64 labels) or 2 (float targets)");
    SGV3D_REQUIRE(workspace_bytes >= sgv3d_focal_loss_workspace_bytes(), "focal_loss: workspace too small");
    SGV3D_REQUIRE(gamma >= 0.f, "focal_loss: gamma must be >= 0");
    SGV3D_REQUIRE(!(use_ignore_index && target_kind == 2), "focal_loss: ignore_index needs integer labels");
    FocalArgs a{};
    a.logit = logits; a.grad = grad; a.target = target;
    a.sb = batch_stride; a.sc = class_stride; a.sp = pixel_stride;
    a.batch = batch; a.classes = num_classes; a.pixels = pixels; a.label_kind = target_kind;
    a.use_alpha = alpha >= 0.f; a.alpha = alpha; a.gamma = gamma;
    a.use_ignore = use_ignore_index; a.ignore_index = ignore_index; a.mean = reduction_mean; a.grad_scale = grad_scale;
    a.partial = static_cast<double *>(workspace); a.out = loss_out;
    hipStream_t s = as_stream(stream);
    if (target_kind != 2) {
        focal_count_kernel<<<kFocalBlocks, kT, 0, s>>>(a);
        if (int rc = check_launch("focal_count_kernel")) return rc;
    }
    focal_kernel<<<kFocalBlocks, kT, 0, s>>>(a);
    if (int rc = check_launch("focal_kernel")) return rc;
    focal_finish_kernel<<<1, kT, 0, s>>>(a);
    return check_launch("focal_finish_kernel");
}

extern "C" int sgv3d_upsample_bilinear2x_backward(int batch, int h, int w, int channels, const float *dy, float *dx,
                                                  void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0, "upsample_bilinear2x_backward: bad shape");
    SGV3D_REQUIRE(dy && dx, "upsample_bilinear2x_backward: null pointer");
    const long long total = (long long)batch * h * w * channels;
    upsample2x_backward_kernel<<<cdiv(total, kT), kT, 0, as_stream(stream)>>>(batch, h, w, channels, dy, dx);
    return check_launch("upsample2x_backward_kernel");
}

extern "C" int sgv3d_add_mul_sigmoid_backward(long long n, const float *dy, const float *b, const float *c, float *db,
                                              float *dc, void *stream) {
    SGV3D_REQUIRE(n > 0 && (n & 3) == 0, "add_mul_sigmoid_backward: n must be a positive multiple of 4");
    SGV3D_REQUIRE(dy && b && c && db && dc, "add_mul_sigmoid_backward: null pointer");
    add_mul_sigmoid_backward_kernel<<<cdiv(n / 4, kT), kT, 0, as_stream(stream)>>>(
        n / 4, reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(b), reinterpret_cast<const float4 *>(c),
        reinterpret_cast<float4 *>(db), reinterpret_cast<float4 *>(dc));
    return check_launch("add_mul_sigmoid_backward_kernel");
}
