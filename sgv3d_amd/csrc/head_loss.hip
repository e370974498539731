// CenterHead detection loss, forward and gradient with respect to the head's output maps, on the device
// (SURVEY.md §8f rank 2).  Reference: BEVHeightHead.loss (layers/heads/bev_height_head.py:255-311) with
// mmdet 2.19.0 GaussianFocalLoss (alpha 2, gamma 4, eps 1e-12) / L1Loss and mmdet3d clip_sigmoid.
// The reference synchronises the host twice per task (.item() of the two averaging factors); here the factors
// stay in a small device tensor ("stats") between the two entry points, so the host can all-reduce them over
// the data-parallel ranks (reduce_mean, bev_height_head.py:271-273,296-297) with one collective and never waits:
//
//   sgv3d_centerhead_loss_stats     stats[0] = number of heatmap cells equal to 1, stats[1] = number of valid
//                                   slots (mask sum), one task per call (exact integer partial counts)
//   sgv3d_centerhead_loss           heat_kernel: clip_sigmoid + Gaussian focal loss + d/dlogit per cell, block
//                                   partial sums; box_kernel: gather of the 10 regression channels at `ind`,
//                                   weighted L1 + its gradient scattered back (cells shared by several boxes
//                                   are summed in slot order by the first of them: deterministic, no float
//                                   atomics); finish_kernel: fixed-order sum of the partials in float64.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kT = 256;
constexpr int kCodes = 10;
constexpr int kStatsBlocks = 256;
constexpr int kHeatBlocks = 1024;

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *lds) {   // kT threads; result valid in thread 0
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kT / 64; ++i) r += lds[i];
    __syncthreads();
    return r;
}

struct StatsArgs {
    const float *target;        // [B][cat][hw] contiguous per sample: base + b*t_stride
    const unsigned char *mask;  // [B][max_objs]
    long long t_stride;
    int batch, cat_hw, max_objs;
    float *stats;               // [2]
    unsigned int *partial;      // workspace: [kStatsBlocks]
};

// Exact integer counts: per-block partials, then one workgroup adds them (and the mask) up.
__global__ void __launch_bounds__(kT) stats_kernel(StatsArgs a) {
    __shared__ unsigned int lds[kT / 64];
    unsigned int pos = 0;
    const long long n = (long long)a.batch * a.cat_hw;
    for (long long e = blockIdx.x * (long long)kT + threadIdx.x; e < n; e += (long long)gridDim.x * kT) {
        const long long b = e / a.cat_hw, r = e - b * a.cat_hw;
        pos += a.target[b * a.t_stride + r] == 1.f ? 1u : 0u;
    }
    const unsigned int total = block_sum<unsigned int>(pos, lds);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kT) stats_finish_kernel(StatsArgs a, int blocks) {
    __shared__ unsigned int lds[kT / 64];
    unsigned int s = 0;
    for (int i = threadIdx.x; i < blocks; i += kT) s += a.partial[i];
    const unsigned int npos = block_sum<unsigned int>(s, lds);
    unsigned int m = 0;
    for (int i = threadIdx.x; i < a.batch * a.max_objs; i += kT) m += a.mask[i] ? 1u : 0u;
    const unsigned int nmask = block_sum<unsigned int>(m, lds);
    if (threadIdx.x == 0) {
        a.stats[0] = (float)npos;
        a.stats[1] = (float)nmask;
    }
}

struct LossArgs {
    const float *heat;       // logits [B][cat][hw], sample stride p_stride
    const float *target;     // [B][cat][hw], sample stride t_stride
    float *g_heat;           // d loss / d logit, addressed like heat (or null)
    const float *maps[5];    // reg, height, dim, rot, vel: [B][c][hw], sample stride p_stride
    float *g_maps[5];        // addressed like maps (or null: no gradients)
    const float *anno;       // [B][max_objs][10]
    const long long *ind;    // [B][max_objs]
    const unsigned char *mask;
    const float *stats;      // [2] (already averaged over the ranks)
    long long p_stride, t_stride, g_stride;
    int batch, cat, hw, max_objs, heat_blocks;
    float code_w[kCodes];
    float box_weight, grad_scale;
    double *partial;         // workspace: [heat_blocks + batch * kBoxSplit]
    float *out;              // [2] = (loss_heatmap, loss_bbox)
};

__global__ void __launch_bounds__(kT) heat_kernel(LossArgs a) {
    __shared__ double lds[kT / 64];
    const float avg = fmaxf(a.stats[0], 1.f);
    const float gs = a.grad_scale / avg;
    const long long n = (long long)a.batch * a.cat * a.hw;
    const long long per = (long long)a.cat * a.hw;
    double acc = 0.0;
    for (long long e = blockIdx.x * (long long)kT + threadIdx.x; e < n; e += (long long)gridDim.x * kT) {
        const long long b = e / per, r = e - b * per;
        const float x = a.heat[b * a.p_stride + r];
        const float t = a.target[b * a.t_stride + r];
        const float s = 1.f / (1.f + expf(-x));
        const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
        const float q = 1.f - p;
        float l, dp;
        if (t == 1.f) {
            const float lg = logf(p + 1e-12f);
            l = -lg * q * q;
            dp = -q * q / (p + 1e-12f) + 2.f * q * lg;
        } else {
            const float u = 1.f - t;
            const float nw = (u * u) * (u * u);
            const float lg = logf(q + 1e-12f);
            l = -lg * p * p * nw;
            dp = (p * p / (q + 1e-12f) - 2.f * p * lg) * nw;
        }
        acc += (double)l;
        if (a.g_heat) {
            const bool inside = s >= 1e-4f && s <= 1.f - 1e-4f;   // clamp passes the gradient on [min, max]
            a.g_heat[b * a.g_stride + r] = inside ? dp * (s * (1.f - s)) * gs : 0.f;
        }
    }
    const double total = block_sum<double>(acc, lds);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = total;
}

// kBoxSplit workgroups per sample, each owning a contiguous range of slots (all of them keep the sample's index
// list in LDS to find the slots that share a cell).
constexpr int kBoxSplit = 8;

__global__ void __launch_bounds__(kT) box_kernel(LossArgs a) {
    extern __shared__ long long s_ind[];   // [max_objs], -1 = masked out
    __shared__ double lds[kT / 64];
    const int b = blockIdx.x / kBoxSplit, part = blockIdx.x % kBoxSplit;
    const float num = fmaxf(a.stats[1], 1e-4f);
    const float gs = a.grad_scale * a.box_weight / num;
    for (int k = threadIdx.x; k < a.max_objs; k += kT) {
        const size_t slot = (size_t)b * a.max_objs + k;
        const long long cell = a.ind[slot];
        s_ind[k] = (a.mask[slot] && cell >= 0 && cell < a.hw) ? cell : -1;   // an index outside the map is dropped
    }
    __syncthreads();
    const int ch_map[kCodes] = {0, 0, 1, 2, 2, 2, 3, 3, 4, 4};
    const int ch_off[kCodes] = {0, 1, 0, 0, 1, 2, 0, 1, 0, 1};
    const int per = (a.max_objs + kBoxSplit - 1) / kBoxSplit;
    const int k_end = min(a.max_objs, (part + 1) * per);
    double acc = 0.0;
    for (int k = part * per + threadIdx.x; k < k_end; k += kT) {
        const long long cell = s_ind[k];
        if (cell < 0) continue;
        const float *tg = a.anno + ((size_t)b * a.max_objs + k) * kCodes;
        float pred[kCodes], g[kCodes];
#pragma unroll
        for (int c = 0; c < kCodes; ++c) {
            pred[c] = a.maps[ch_map[c]][(size_t)b * a.p_stride + (size_t)ch_off[c] * a.hw + cell];
            const float t = tg[c];
            const float w = (t == t) ? a.code_w[c] : 0.f;      // isnotnan mask (bev_height_head.py:298-299)
            acc += (double)(fabsf(pred[c] - t) * w);
            g[c] = 0.f;
        }
        if (!a.g_maps[0]) continue;
        bool first = true;
        for (int j = 0; j < k; ++j) first = first && s_ind[j] != cell;
        if (!first) continue;
        // gather backward = scatter-add: the first slot on a cell adds up every slot on it, in slot order
        for (int j = k; j < a.max_objs; ++j) {
            if (s_ind[j] != cell) continue;
            const float *tj = a.anno + ((size_t)b * a.max_objs + j) * kCodes;
#pragma unroll
            for (int c = 0; c < kCodes; ++c) {
                const float t = tj[c];
                const float w = (t == t) ? a.code_w[c] : 0.f;
                const float d = pred[c] - t;
                g[c] += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
        }
#pragma unroll
        for (int c = 0; c < kCodes; ++c)
            a.g_maps[ch_map[c]][(size_t)b * a.g_stride + (size_t)ch_off[c] * a.hw + cell] = g[c] * gs;
    }
    const double total = block_sum<double>(acc, lds);
    if (threadIdx.x == 0) a.partial[a.heat_blocks + blockIdx.x] = total;
}

__global__ void __launch_bounds__(kT) finish_kernel(LossArgs a) {
    __shared__ double lds[kT / 64];
    double h = 0.0, x = 0.0;
    // fixed assignment of partials to threads and fixed tree: the same sum on every run
    for (int i = threadIdx.x; i < a.heat_blocks; i += kT) h += a.partial[i];
    for (int i = threadIdx.x; i < a.batch * kBoxSplit; i += kT) x += a.partial[a.heat_blocks + i];
    const double hs = block_sum<double>(h, lds);
    const double xs = block_sum<double>(x, lds);
    if (threadIdx.x == 0) {
        a.out[0] = (float)(hs / (double)fmaxf(a.stats[0], 1.f));
        a.out[1] = (float)(xs / (double)fmaxf(a.stats[1], 1e-4f) * (double)a.box_weight);
    }
}

__global__ void __launch_bounds__(kT) zero_maps_kernel(LossArgs a) {
    const int c_of[5] = {2, 1, 3, 2, 2};
    for (int m = 0; m < 5; ++m) {
        float *g = a.g_maps[m];
        if (!g) continue;
        const long long per = (long long)c_of[m] * a.hw, n = per * a.batch;
        for (long long e = blockIdx.x * (long long)kT + threadIdx.x; e < n; e += (long long)gridDim.x * kT) {
            const long long b = e / per, r = e - b * per;
            g[b * a.g_stride + r] = 0.f;
        }
    }
}

}  // namespace

extern "C" size_t sgv3d_centerhead_loss_workspace_bytes(int batch) {
    return 16 + (size_t)kStatsBlocks * 4 + ((size_t)kHeatBlocks + (size_t)(batch > 0 ? batch : 0) * 8) * 8 + 16;
}

extern "C" int sgv3d_centerhead_loss_stats(int batch, int num_class, int h, int w, int max_objs,
                                           const float *target_heatmap, long long target_batch_stride,
                                           const unsigned char *mask, float *stats, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && num_class > 0 && h > 0 && w > 0 && max_objs > 0, "centerhead_loss_stats: bad sizes");
    SGV3D_REQUIRE(target_heatmap && mask && stats && workspace, "centerhead_loss_stats: null pointer");
    SGV3D_REQUIRE(workspace_bytes >= sgv3d_centerhead_loss_workspace_bytes(batch), "centerhead_loss_stats: workspace too small");
    SGV3D_REQUIRE(target_batch_stride >= (long long)num_class * h * w, "centerhead_loss_stats: bad stride");
    StatsArgs a{};
    a.target = target_heatmap; a.mask = mask; a.t_stride = target_batch_stride; a.batch = batch;
    a.cat_hw = num_class * h * w; a.max_objs = max_objs; a.stats = stats;
    a.partial = static_cast<unsigned int *>(workspace) + 4;
    hipStream_t s = as_stream(stream);
    stats_kernel<<<kStatsBlocks, kT, 0, s>>>(a);
    if (int rc = check_launch("stats_kernel")) return rc;
    stats_finish_kernel<<<1, kT, 0, s>>>(a, kStatsBlocks);
    return check_launch("stats_finish_kernel");
}

extern "C" int sgv3d_centerhead_loss(int batch, int num_class, int h, int w, int max_objs, const float *heatmap,
                                     const float *reg, const float *height, const float *dim, const float *rot,
                                     const float *vel, long long pred_batch_stride, const float *target_heatmap,
                                     long long target_batch_stride, const float *anno_box, const long long *ind,
                                     const unsigned char *mask, const float *stats, const float *code_weights,
                                     float loss_bbox_weight, float grad_scale, float *g_heatmap, float *g_reg,
                                     float *g_height, float *g_dim, float *g_rot, float *g_vel,
                                     long long grad_batch_stride, float *loss_out,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && num_class > 0 && h > 0 && w > 0 && max_objs > 0, "centerhead_loss: bad sizes");
    SGV3D_REQUIRE(heatmap && reg && height && dim && rot && vel, "centerhead_loss: null prediction map");
    SGV3D_REQUIRE(target_heatmap && anno_box && ind && mask && stats && code_weights && loss_out && workspace,
                  "centerhead_loss: null pointer");
    SGV3D_REQUIRE(workspace_bytes >= sgv3d_centerhead_loss_workspace_bytes(batch), "centerhead_loss: workspace too small");
    const bool grads = g_heatmap != nullptr;
    SGV3D_REQUIRE(grads == (g_reg && g_height && g_dim && g_rot && g_vel) || (!grads && !g_reg && !g_height && !g_dim && !g_rot && !g_vel),
                  "centerhead_loss: pass all six gradient maps or none");
    SGV3D_REQUIRE((size_t)max_objs * 8 <= 64 * 1024, "centerhead_loss: max_objs > 8192");
    LossArgs a{};
    a.heat = heatmap; a.target = target_heatmap; a.g_heat = g_heatmap;
    const float *maps[5] = {reg, height, dim, rot, vel};
    float *gm[5] = {g_reg, g_height, g_dim, g_rot, g_vel};
    for (int i = 0; i < 5; ++i) { a.maps[i] = maps[i]; a.g_maps[i] = gm[i]; }
    a.anno = anno_box; a.ind = ind; a.mask = mask; a.stats = stats;
    a.p_stride = pred_batch_stride; a.t_stride = target_batch_stride; a.g_stride = grad_batch_stride;
    a.batch = batch; a.cat = num_class; a.hw = h * w; a.max_objs = max_objs; a.heat_blocks = kHeatBlocks;
    for (int c = 0; c < kCodes; ++c) a.code_w[c] = code_weights[c];
    a.box_weight = loss_bbox_weight; a.grad_scale = grad_scale;
    a.partial = reinterpret_cast<double *>(static_cast<char *>(workspace) + 16 + (size_t)kStatsBlocks * 4);
    a.out = loss_out;
    hipStream_t s = as_stream(stream);
    if (grads) {
        zero_maps_kernel<<<512, kT, 0, s>>>(a);
        if (int rc = check_launch("zero_maps_kernel")) return rc;
    }
    heat_kernel<<<kHeatBlocks, kT, 0, s>>>(a);
    if (int rc = check_launch("heat_kernel")) return rc;
    box_kernel<<<batch * kBoxSplit, kT, (size_t)max_objs * 8, s>>>(a);
    if (int rc = check_launch("box_kernel")) return rc;
    finish_kernel<<<1, kT, 0, s>>>(a);
    return check_launch("finish_kernel");
}
