// Training-mode kernels of the layers that ran as ATen operators in round 1:
//
//   sgv3d_maxpool3x3s2_train_forward / _backward   nn.MaxPool2d(3, 2, 1) of the image ResNet stem (mmdet ResNet, built at
//        layers/backbones/lss_fpn.py:296) with the arg-max tap kept as one byte per output, and its adjoint in GATHER form:
//        every input element sums the (at most four) windows that selected it -- deterministic, no atomics.  Ties go to the
//        first maximum in row-major window order, as torch's kernel does.
//   sgv3d_dense_backward_weight   dW[n][k] = sum_b dY[b][n] X[b][k]: weight gradient of the ASPP image-pooling branch's
//        1x1 convolution, which acts on a [B, C] vector (lss_fpn.py:80-88,101); its forward and data gradient are the
//        `dense` kernel.
//   sgv3d_deform_im2col3x3_backward   adjoint of the deformable bilinear im2col of mmcv's DeformConv2dPack
//        (lss_fpn.py:190-198): d loss / d offsets (one wave per (pixel, tap), channels reduced in fixed order) and
//        d loss / d input (scatter to the four corners with float atomics, like mmcv's deformable_col2im).
//
// All HBM-bound streaming work: one read of the operands, one write of the results.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kT = 256;

// ------------------------------------------------------------------------------------------------ max pooling
__global__ __launch_bounds__(kT) void maxpool_train_fwd_kernel(int B, int H, int W, int C4, int OH, int OW,
                                                              const float4 *__restrict__ x, float4 *__restrict__ y,
                                                              uchar4 *__restrict__ idx) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    const long long total = (long long)B * OH * OW * C4;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned char am[4] = {255, 255, 255, 255};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh * 2 - 1 + dy;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iw = ow * 2 - 1 + dx;
            if ((unsigned)iw >= (unsigned)W) continue;
            const float4 v4 = x[((long long)(b * H + ih) * W + iw) * C4 + c];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (am[j] == 255 || v[j] > m[j] || v[j] != v[j]) {      // first maximum wins; NaN propagates like torch
                    m[j] = v[j];
                    am[j] = (unsigned char)(dy * 3 + dx);
                }
        }
    }
    y[i] = make_float4(m[0], m[1], m[2], m[3]);
    idx[i] = make_uchar4(am[0], am[1], am[2], am[3]);
}

__global__ __launch_bounds__(kT) void maxpool_bwd_kernel(int B, int H, int W, int C4, int OH, int OW,
                                                        const uchar4 *__restrict__ idx, const float4 *__restrict__ dy,
                                                        float4 *__restrict__ dx) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    const long long total = (long long)B * H * W * C4;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int b = (int)(t / H);
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    // windows containing (ih, iw): oh in {ih / 2, (ih + 1) / 2}, ascending -- fixed order of the (<= 4) terms
    const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1, ow0 = iw >> 1, ow1 = (iw + 1) >> 1;
    for (int a = 0; a < 2; ++a) {
        const int oh = a == 0 ? oh0 : oh1;
        if ((a == 1 && oh1 == oh0) || oh >= OH) continue;
        const int ty = ih - (oh * 2 - 1);
        for (int e = 0; e < 2; ++e) {
            const int ow = e == 0 ? ow0 : ow1;
            if ((e == 1 && ow1 == ow0) || ow >= OW) continue;
            const int tap = ty * 3 + (iw - (ow * 2 - 1));
            const long long o = ((long long)(b * OH + oh) * OW + ow) * C4 + c;
            const uchar4 k = idx[o];
            const float4 d = dy[o];
            g[0] += k.x == tap ? d.x : 0.f;
            g[1] += k.y == tap ? d.y : 0.f;
            g[2] += k.z == tap ? d.z : 0.f;
            g[3] += k.w == tap ? d.w : 0.f;
        }
    }
    dx[i] = make_float4(g[0], g[1], g[2], g[3]);
}

// ------------------------------------------------------------------------------------------------ dense: weight gradient
__global__ __launch_bounds__(kT) void dense_bwd_weight_kernel(int B, int K, int N, const float *__restrict__ x,
                                                             const float *__restrict__ dy, float *__restrict__ dw) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long long)n * K);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * N + n] * x[(size_t)b * K + k];
    dw[i] = s;
}

// ------------------------------------------------------------------------------------------------ deformable im2col adjoint
// one wave per (pixel, tap); lanes walk consecutive channels.  dcol uses the layout of sgv3d_deform_im2col3x3:
// [pixel][group][tap][channels per group].
__global__ __launch_bounds__(kT) void deform_im2col_bwd_kernel(int B, int H, int W, int C, int groups,
                                                              const float *__restrict__ x, const float *__restrict__ off,
                                                              int off_ld, const float *__restrict__ dcol,
                                                              float *__restrict__ dx, float *__restrict__ doff, int doff_ld) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
    const long long total = (long long)B * H * W * 9;
    if (item >= total) return;                                   // wave-uniform
    const int tap = (int)(item % 9);
    const long long p = item / 9;
    const int w_ = (int)(p % W);
    const long long t2 = p / W;
    const int h_ = (int)(t2 % H);
    const int b = (int)(t2 / H);
    const int ky = tap / 3, kx = tap - ky * 3;
    const float oy = off[p * off_ld + 2 * tap], ox = off[p * off_ld + 2 * tap + 1];
    const float hf = (float)(h_ - 1 + ky) + oy;
    const float wf = (float)(w_ - 1 + kx) + ox;
    float gy = 0.f, gx = 0.f;
    if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {
        const int h_low = (int)floorf(hf), w_low = (int)floorf(wf);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = hf - (float)h_low, lw = wf - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
        const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const long long img = (long long)b * H * W;
        const long long o1 = (img + (long long)h_low * W + w_low) * C, o2 = (img + (long long)h_low * W + w_high) * C;
        const long long o3 = (img + (long long)h_high * W + w_low) * C, o4 = (img + (long long)h_high * W + w_high) * C;
        const int cpg = C / groups;
        // lanes walk CONSECUTIVE channels: every load and every atomic instruction of the wave covers one contiguous 256-byte
        // piece of a pixel row (with a float4 per lane the four atomics of a lane were 16 bytes apart from the next lane's:
        // 2.35 ms for the cfg-2 layer at batch 2; the memory-side atomic units take whole 64-byte requests)
        for (int c = lane; c < C; c += 64) {
            const int g = c / cpg, cg = c - g * cpg;
            const float d = dcol[((p * groups + g) * 9 + tap) * cpg + cg];
            const float v1 = ok1 ? x[o1 + c] : 0.f, v2 = ok2 ? x[o2 + c] : 0.f;
            const float v3 = ok3 ? x[o3 + c] : 0.f, v4 = ok4 ? x[o4 + c] : 0.f;
            // d val / d hf = hw (v3 - v1) + lw (v4 - v2);  d val / d wf = hh (v2 - v1) + lh (v4 - v3)
            gy += d * (hw * (v3 - v1) + lw * (v4 - v2));
            gx += d * (hh * (v2 - v1) + lh * (v4 - v3));
            if (ok1) __hip_atomic_fetch_add(dx + o1 + c, w1 * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok2) __hip_atomic_fetch_add(dx + o2 + c, w2 * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok3) __hip_atomic_fetch_add(dx + o3 + c, w3 * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok4) __hip_atomic_fetch_add(dx + o4 + c, w4 * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {                            // butterfly: the same order for every lane, every run
        gy += __shfl_xor(gy, o, 64);
        gx += __shfl_xor(gx, o, 64);
    }
    if (lane == 0) {
        doff[p * doff_ld + 2 * tap] = gy;
        doff[p * doff_ld + 2 * tap + 1] = gx;
    }
}

__global__ __launch_bounds__(kT) void zero_f32_kernel(long long n4, float4 *__restrict__ p) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// out[ci][co][t] = w[co][ci][taps - 1 - t]: the weights of the data-gradient convolution (flipped taps, in / out swapped)
__global__ __launch_bounds__(kT) void weight_rot180_transpose_kernel(const float *__restrict__ w, int cout, int cin, int taps,
                                                                     float *__restrict__ out) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= cout * cin * taps) return;
    const int t = i % taps, r = i / taps;
    const int co = r % cout, ci = r / cout;
    out[i] = w[((size_t)co * cin + ci) * taps + (taps - 1 - t)];
}

// ------------------------------------------------------------------------------------------------ packed weights, all layers
// One launch refreshes every packed weight form of a model after an optimiser step: every packed form the training step reads
// (implicit-GEMM rows, rotated / transposed data-gradient rows, bf16 fragment orders) is a PERMUTATION of a parameter tensor with
// zero padding, so a job is (parameter, packed buffer, index map); the maps are made once by running the layer's own pack kernels on
// an index-valued weight tensor (sgv3d_amd/pack_cache.py).  Replaces ~270 pack / rotate / flip launches of 5-10 us per step.
struct GatherJob {
    const float *src;        // the parameter (flat)
    void *dst;               // packed buffer: f32 or bf16
    const int *idx;          // [n]: element of src, or -1 = 0
    long long n;
    int first_block;         // blocks [first_block, next job's first_block) work on this job
    int bf16;                // dst dtype
};
constexpr int kGatherPerBlock = kT * 8;

__global__ __launch_bounds__(kT) void gather_pack_kernel(const GatherJob *__restrict__ jobs, int n_jobs) {
    // the job of this block: the last one whose first_block <= blockIdx.x
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const GatherJob j = jobs[lo];
    const long long base = (long long)((int)blockIdx.x - j.first_block) * kGatherPerBlock;
    // two 16-byte index loads per thread, issued before the gathers
    const long long i0 = base + (long long)threadIdx.x * 4, i1 = i0 + kT * 4;
    int4 a = make_int4(-1, -1, -1, -1), b = a;
    const bool full0 = i0 + 4 <= j.n, full1 = i1 + 4 <= j.n;
    if (full0) a = *reinterpret_cast<const int4 *>(j.idx + i0);
    else if (i0 < j.n) { a.x = j.idx[i0]; if (i0 + 1 < j.n) a.y = j.idx[i0 + 1]; if (i0 + 2 < j.n) a.z = j.idx[i0 + 2]; }
    if (full1) b = *reinterpret_cast<const int4 *>(j.idx + i1);
    else if (i1 < j.n) { b.x = j.idx[i1]; if (i1 + 1 < j.n) b.y = j.idx[i1 + 1]; if (i1 + 2 < j.n) b.z = j.idx[i1 + 2]; }
    auto get = [&](int k) { return k >= 0 ? j.src[k] : 0.f; };
    const float v[8] = {get(a.x), get(a.y), get(a.z), get(a.w), get(b.x), get(b.y), get(b.z), get(b.w)};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long i = h ? i1 : i0;
        if (i >= j.n) continue;
        const bool full = i + 4 <= j.n;
        if (j.bf16) {
            __bf16 *d = static_cast<__bf16 *>(j.dst) + i;
            if (full) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<bf16x4 *>(d) = bf16x4{(__bf16)v[4 * h], (__bf16)v[4 * h + 1], (__bf16)v[4 * h + 2], (__bf16)v[4 * h + 3]};
            } else {
                for (int u = 0; u < 4 && i + u < j.n; ++u) d[u] = (__bf16)v[4 * h + u];
            }
        } else {
            float *d = static_cast<float *>(j.dst) + i;
            if (full) *reinterpret_cast<float4 *>(d) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            else for (int u = 0; u < 4 && i + u < j.n; ++u) d[u] = v[4 * h + u];
        }
    }
}

}  // namespace

extern "C" int sgv3d_gather_pack_job_bytes(void) { return (int)sizeof(GatherJob); }
extern "C" int sgv3d_gather_pack_elements_per_block(void) { return kGatherPerBlock; }

extern "C" int sgv3d_gather_pack(const void *jobs, int n_jobs, int total_blocks, void *stream) {
    SGV3D_REQUIRE(jobs && n_jobs > 0 && total_blocks > 0, "gather_pack: bad argument");
    hipLaunchKernelGGL(gather_pack_kernel, dim3(total_blocks), dim3(kT), 0, as_stream(stream), static_cast<const GatherJob *>(jobs), n_jobs);
    return check_launch("gather_pack_kernel");
}

extern "C" int sgv3d_weight_rot180_transpose(const float *w, int cout, int cin, int kh, int kw, float *out, void *stream) {
    SGV3D_REQUIRE(w && out && cout > 0 && cin > 0 && kh > 0 && kw > 0 && (long long)cout * cin * kh * kw < 0x7fffffffLL,
                  "weight_rot180_transpose: bad argument");
    hipLaunchKernelGGL(weight_rot180_transpose_kernel, dim3(cdiv((long long)cout * cin * kh * kw, kT)), dim3(kT), 0, as_stream(stream),
                       w, cout, cin, kh * kw, out);
    return check_launch("weight_rot180_transpose_kernel");
}

extern "C" int sgv3d_maxpool3x3s2_train_forward(int batch, int in_h, int in_w, int channels, const float *x, float *y,
                                                unsigned char *argmax, void *stream) {
    SGV3D_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && channels > 0 && channels % 4 == 0, "maxpool3x3s2_train_forward: bad shape");
    SGV3D_REQUIRE(x && y && argmax, "maxpool3x3s2_train_forward: null pointer");
    const int oh = (in_h - 1) / 2 + 1, ow = (in_w - 1) / 2 + 1;
    const long long total = (long long)batch * oh * ow * (channels / 4);
    hipLaunchKernelGGL(maxpool_train_fwd_kernel, dim3(cdiv(total, kT)), dim3(kT), 0, as_stream(stream), batch, in_h, in_w,
                       channels / 4, oh, ow, reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y),
                       reinterpret_cast<uchar4 *>(argmax));
    return check_launch("maxpool_train_fwd_kernel");
}

extern "C" int sgv3d_maxpool3x3s2_backward(int batch, int in_h, int in_w, int channels, const unsigned char *argmax,
                                           const float *grad_out, float *grad_in, void *stream) {
    SGV3D_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && channels > 0 && channels % 4 == 0, "maxpool3x3s2_backward: bad shape");
    SGV3D_REQUIRE(argmax && grad_out && grad_in, "maxpool3x3s2_backward: null pointer");
    const int oh = (in_h - 1) / 2 + 1, ow = (in_w - 1) / 2 + 1;
    const long long total = (long long)batch * in_h * in_w * (channels / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(cdiv(total, kT)), dim3(kT), 0, as_stream(stream), batch, in_h, in_w, channels / 4,
                       oh, ow, reinterpret_cast<const uchar4 *>(argmax), reinterpret_cast<const float4 *>(grad_out),
                       reinterpret_cast<float4 *>(grad_in));
    return check_launch("maxpool_bwd_kernel");
}

extern "C" int sgv3d_dense_backward_weight(int batch, int k, int n, const float *x, const float *grad_out, float *grad_w,
                                           void *stream) {
    SGV3D_REQUIRE(batch > 0 && k > 0 && n > 0 && x && grad_out && grad_w, "dense_backward_weight: bad argument");
    hipLaunchKernelGGL(dense_bwd_weight_kernel, dim3(cdiv((long long)n * k, kT)), dim3(kT), 0, as_stream(stream), batch, k, n, x,
                       grad_out, grad_w);
    return check_launch("dense_bwd_weight_kernel");
}

extern "C" int sgv3d_deform_im2col3x3_backward(int batch, int h, int w, int channels, int groups, const float *x,
                                               const float *offset, int off_ld, const float *grad_col, float *grad_x,
                                               float *grad_offset, int grad_off_ld, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0 && groups > 0 && channels % (4 * groups) == 0 && off_ld >= 18 &&
                      grad_off_ld >= 18,
                  "deform_im2col3x3_backward: bad shape");
    SGV3D_REQUIRE(x && offset && grad_col && grad_x && grad_offset, "deform_im2col3x3_backward: null pointer");
    const long long n4 = (long long)batch * h * w * channels / 4;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(cdiv(n4, kT)), dim3(kT), 0, as_stream(stream), n4, reinterpret_cast<float4 *>(grad_x));
    const long long items = (long long)batch * h * w * 9;
    hipLaunchKernelGGL(deform_im2col_bwd_kernel, dim3(cdiv(items, kT / 64)), dim3(kT), 0, as_stream(stream), batch, h, w, channels,
                       groups, x, offset, off_ld, grad_col, grad_x, grad_offset, grad_off_ld);
    return check_launch("deform_im2col_bwd_kernel");
}
