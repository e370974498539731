// CenterPoint box decode + circle NMS on the device.
// Reference: BEVHeight.get_bboxes (models/bev_height.py:116-126) -> mmdet3d 0.18.1
// CenterHead.get_bboxes / CenterPointBBoxCoder.decode / circle_nms (source not in the reference
// repository: restated from the published algorithm; the reference runs the NMS
// on the CPU through numba with a device->host copy per task).
//
// Per task (one C-ABI call, three launches, no host synchronisation):
//   1. topk_per_class   sigmoid + exact top-K per (sample, class) over H*W: 4-pass MSB radix select on
//                       order-preserving float keys, deterministic compaction in index order, bitonic
//                       sort by (score desc, index asc) in LDS
//   2. merge_decode     top-K over classes x K, gather of the regression maps, box assembly
//                       (x = (col + reg_x) * out_size_factor * voxel + pc_range, dim = exp, rot = atan2),
//                       score / centre-range mask
//   3. circle_nms       greedy suppression by squared centre distance in score order, one workgroup
//                       per sample, the inner "suppress everything after i" loop in parallel
// Ties (equal scores) are broken by the lower flat index; the reference leaves them to torch.topk /
// numpy argsort, i.e. unspecified.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kTk = 1024;     // threads of the top-k workgroup
constexpr int kMaxK = 1024;   // K padded to a power of two must fit the LDS sort buffers

__device__ __forceinline__ unsigned fkey(float f) {   // ascending order-preserving key
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// (score desc, index asc): a goes before b
__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia < ib); }

// In-LDS bitonic sort of n (power of two) (score, index) pairs into "before" order by T threads.
template <int T>
__device__ void bitonic_pairs(float *sc, int *ix, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += T) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const float sa = sc[i], sb = sc[p];
                    const int ia = ix[i], ib = ix[p];
                    const bool ok = before(sa, ia, sb, ib);      // already in "before" order
                    if (up ? !ok : ok) { sc[i] = sb; sc[p] = sa; ix[i] = ib; ix[p] = ia; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ float fkey_inv(unsigned k) {   // the float a key came from
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// CACHE: the thread's slice is at most kKeyCache elements (H*W <= 65536): the sigmoid keys are computed ONCE and kept in
// registers for the four radix passes and the two compaction passes (the plain form evaluates expf six times per element:
// ~90 us per class at 256 x 256 on a grid of one or two workgroups, i.e. pure latency on the harness's critical path).
constexpr int kKeyCache = 64;

// All tasks of the head in ONE launch per stage (blockIdx.y = task): the six tasks of the shipped configs used to be 18
// dependent launches of one or two workgroups each -- pure latency on the harness's critical path.
constexpr int kMaxTasks = 16;
struct DecodeTasks {
    const float *heat[kMaxTasks], *reg[kMaxTasks], *hei[kMaxTasks], *dim[kMaxTasks], *rot[kMaxTasks], *vel[kMaxTasks];
    int cat[kMaxTasks];
    float nms[kMaxTasks];
};

template <bool CACHE>
__global__ __launch_bounds__(kTk) void topk_per_class_kernel(int batch, int max_cat, long long hw, int K, int Kp,
                                                             const DecodeTasks ts, long long batch_stride,
                                                             float *__restrict__ out_score, int *__restrict__ out_ind) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_krem;
    __shared__ int s_cnt[kTk / 64][2];
    __shared__ float ssc[kMaxK];
    __shared__ int six[kMaxK];
    const int task = blockIdx.y, cat = ts.cat[task];
    const int b = blockIdx.x / max_cat, c = blockIdx.x - b * max_cat;
    if (c >= cat) return;                                   // (block-uniform)
    const float *src = ts.heat[task] + (long long)b * batch_stride + (long long)c * hw;
    const long long slot = ((long long)task * batch + b) * max_cat + c;      // this (task, sample, class)'s K outputs
    const int tid = threadIdx.x;
    const long long per = (hw + kTk - 1) / kTk;            // contiguous slice per thread (index order)
    const long long i0 = tid * per, i1 = min(hw, i0 + per);
    auto score = [&](long long i) { return 1.f / (1.f + expf(-src[i])); };
    unsigned keys[CACHE ? kKeyCache : 1];
    const int cnt = (int)(i1 > i0 ? i1 - i0 : 0);           // CACHE: elements of this thread's slice (<= kKeyCache), 32-bit from here on
    const int base32 = (int)i0;
    const float *src32 = src + i0;
    if constexpr (CACHE) {
        // a thread's slice is contiguous (the compaction below keeps index order): 16-byte loads where the slice allows them --
        // a scalar load of a wave touches 64 cache lines for 256 useful bytes, and there would be 64 of them per thread
        const bool vec = (per & 3) == 0 && (hw & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;     // block-uniform
        if (vec) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src32);
#pragma unroll
            for (int q4 = 0; q4 < kKeyCache / 4; ++q4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool in = 4 * q4 < cnt;
                if (in) v = s4[q4];
                keys[4 * q4 + 0] = in ? fkey(1.f / (1.f + expf(-v.x))) : 0u;          // (0 < every sigmoid key)
                keys[4 * q4 + 1] = in ? fkey(1.f / (1.f + expf(-v.y))) : 0u;
                keys[4 * q4 + 2] = in ? fkey(1.f / (1.f + expf(-v.z))) : 0u;
                keys[4 * q4 + 3] = in ? fkey(1.f / (1.f + expf(-v.w))) : 0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kKeyCache; ++q) keys[q] = (q < cnt) ? fkey(1.f / (1.f + expf(-src32[q]))) : 0u;
        }
    }
    // ---- radix select of the K-th largest key ------------------------------------------------
    if (tid == 0) { s_prefix = 0; s_krem = (unsigned)K; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        if constexpr (CACHE) {
            // Heat maps are smooth: neighbouring cells share the leading digits of their keys, and ALL lanes of a wave would hit
            // the same histogram bin -- an LDS atomic on one address is serialised, 65 536 of them per pass took ~30 us.  Each
            // lane therefore counts runs of equal digits along its slice and adds a whole run at once.
            unsigned cur = 0u, run = 0u;
#pragma unroll
            for (int q = 0; q < kKeyCache; ++q) {
                const unsigned k = keys[q];
                if (q < cnt && (k & himask) == (prefix & himask)) {
                    const unsigned d = (k >> shift) & 255u;
                    if (run != 0u && d != cur) { atomicAdd(&hist[cur], run); run = 0u; }
                    cur = d;
                    ++run;
                }
            }
            if (run != 0u) atomicAdd(&hist[cur], run);
        } else {
            for (long long i = i0; i < i1; ++i) {
                const unsigned k = fkey(score(i));
                if ((k & himask) == (prefix & himask)) atomicAdd(&hist[(k >> shift) & 255], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0, krem = s_krem;
            int d = 255;
            for (; d > 0; --d) {
                if (cum + hist[d] >= krem) break;
                cum += hist[d];
            }
            s_prefix = prefix | ((unsigned)d << shift);
            s_krem = krem - cum;       // how many are still needed among keys with this digit
        }
        __syncthreads();
    }
    const unsigned kth = s_prefix;
    const int need_eq = (int)s_krem;                        // ties at the K-th key to take (lowest index first)
    // ---- deterministic compaction in index order ---------------------------------------------
    int n_gt = 0, n_eq = 0;
    if constexpr (CACHE) {
#pragma unroll
        for (int q = 0; q < kKeyCache; ++q) {
            const bool in = q < cnt;
            n_gt += in && keys[q] > kth;
            n_eq += in && keys[q] == kth;
        }
    } else {
        for (long long i = i0; i < i1; ++i) {
            const unsigned k = fkey(score(i));
            n_gt += k > kth;
            n_eq += k == kth;
        }
    }
    // block exclusive scans of (n_gt, n_eq)
    int inc_gt = n_gt, inc_eq = n_eq;
    const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(inc_gt, d, 64), e = __shfl_up(inc_eq, d, 64);
        if (lane >= d) { inc_gt += a; inc_eq += e; }
    }
    if (lane == 63) { s_cnt[wid][0] = inc_gt; s_cnt[wid][1] = inc_eq; }
    __syncthreads();
    int base_gt = 0, base_eq = 0, tot_gt = 0;
    for (int w = 0; w < kTk / 64; ++w) {
        if (w < wid) { base_gt += s_cnt[w][0]; base_eq += s_cnt[w][1]; }
        tot_gt += s_cnt[w][0];
    }
    int pos_gt = base_gt + inc_gt - n_gt, pos_eq = base_eq + inc_eq - n_eq;
    for (int i = tid; i < Kp; i += kTk) { ssc[i] = -INFINITY; six[i] = 0x7fffffff; }
    __syncthreads();
    if constexpr (CACHE) {
#pragma unroll
        for (int q = 0; q < kKeyCache; ++q) {
            const unsigned k = keys[q];
            if (q < cnt) {
                const float s = fkey_inv(k);
                if (k > kth) { ssc[pos_gt] = s; six[pos_gt] = base32 + q; ++pos_gt; }
                else if (k == kth) { if (pos_eq < need_eq) { ssc[tot_gt + pos_eq] = s; six[tot_gt + pos_eq] = base32 + q; } ++pos_eq; }
            }
        }
    } else {
        for (long long i = i0; i < i1; ++i) {
            const float s = score(i);
            const unsigned k = fkey(s);
            if (k > kth) { ssc[pos_gt] = s; six[pos_gt] = (int)i; ++pos_gt; }
            else if (k == kth) { if (pos_eq < need_eq) { ssc[tot_gt + pos_eq] = s; six[tot_gt + pos_eq] = (int)i; } ++pos_eq; }
        }
    }
    __syncthreads();
    bitonic_pairs<kTk>(ssc, six, Kp);
    for (int i = tid; i < K; i += kTk) {
        out_score[slot * K + i] = ssc[i];
        out_ind[slot * K + i] = six[i];
    }
}

struct DecodeCfg {
    float out_size_factor, vx, vy, pcx, pcy, score_thr;
    float range[6];
    int norm_bbox, has_range, has_vel;
};

__global__ __launch_bounds__(kTk) void merge_decode_kernel(int batch, int max_cat, int h, int w, int K, int Kp2max,
                                                           const float *__restrict__ cls_score_all, const int *__restrict__ cls_ind_all,
                                                           const DecodeTasks ts, long long batch_stride,
                                                           DecodeCfg cfg, float *__restrict__ boxes, float *__restrict__ scores,
                                                           int *__restrict__ labels, unsigned char *__restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    float *ssc = reinterpret_cast<float *>(dsm);
    int *six = reinterpret_cast<int *>(dsm + sizeof(float) * Kp2max);
    const int task = blockIdx.y, cat = ts.cat[task];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long hw = (long long)h * w;
    const int n = cat * K;
    int Kp2 = 1;
    while (Kp2 < n) Kp2 <<= 1;                               // this task's sort size (<= Kp2max)
    // (the top-k stage stored class c of (task, sample) at rows (task * batch + b) * max_cat + c: contiguous over the classes)
    const float *cls_score = cls_score_all + ((long long)task * batch + b) * max_cat * K;
    const int *cls_ind = cls_ind_all + ((long long)task * batch + b) * max_cat * K;
    const float *reg = ts.reg[task], *hei = ts.hei[task], *dim = ts.dim[task], *rot = ts.rot[task], *vel = ts.vel[task];
    const bool has_vel = vel != nullptr;
    for (int i = tid; i < Kp2; i += kTk) {
        if (i < n) { ssc[i] = cls_score[i]; six[i] = i; }    // flat index = class*K + rank
        else { ssc[i] = -INFINITY; six[i] = 0x7fffffff; }
    }
    __syncthreads();
    bitonic_pairs<kTk>(ssc, six, Kp2);
    for (int t = tid; t < K; t += kTk) {
        const float s = ssc[t];
        const int flat = six[t];
        const long long o = ((long long)task * batch + b) * K + t;
        if (flat >= n || s == -INFINITY) {                       // fewer than K candidates
            for (int q = 0; q < 9; ++q) boxes[o * 9 + q] = 0.f;
            scores[o] = 0.f; labels[o] = 0; valid[o] = 0;
            continue;
        }
        const int cls = flat / K;
        const int ind = cls_ind[flat];
        const int yq = (int)((float)ind / (float)w);            // topk_ys: (ind.float() / width).int()
        const int xq = ind % w;
        const long long bo = (long long)b * batch_stride + ind;
        const float xs = ((float)xq + reg[bo]) * cfg.out_size_factor * cfg.vx + cfg.pcx;
        const float ys = ((float)yq + reg[bo + hw]) * cfg.out_size_factor * cfg.vy + cfg.pcy;
        const float z = hei[bo];
        float d0 = dim[bo], d1 = dim[bo + hw], d2 = dim[bo + 2 * hw];
        if (cfg.norm_bbox) { d0 = expf(d0); d1 = expf(d1); d2 = expf(d2); }
        const float r = atan2f(rot[bo], rot[bo + hw]);          // atan2(sin, cos)
        float *bx = boxes + o * 9;
        bx[0] = xs; bx[1] = ys; bx[2] = z; bx[3] = d0; bx[4] = d1; bx[5] = d2; bx[6] = r;
        bx[7] = has_vel ? vel[bo] : 0.f;
        bx[8] = has_vel ? vel[bo + hw] : 0.f;
        bool ok = s > cfg.score_thr;
        if (cfg.has_range)
            ok = ok && xs >= cfg.range[0] && ys >= cfg.range[1] && z >= cfg.range[2] && xs <= cfg.range[3] &&
                 ys <= cfg.range[4] && z <= cfg.range[5];
        scores[o] = s;
        labels[o] = cls;
        valid[o] = ok ? 1 : 0;
    }
}

// greedy circle NMS over the valid candidates of one sample, in score order (they are sorted already): the plain form, two
// workgroup barriers per candidate (K > kNmsMaskK only; the shipped configs have K = 500)
__global__ __launch_bounds__(512) void circle_nms_serial_kernel(int K, const float *__restrict__ boxes,
                                                         const unsigned char *__restrict__ valid, const DecodeTasks ts,
                                                         int post_max_size, unsigned char *__restrict__ keep) {
    const float thresh = ts.nms[blockIdx.y];
    boxes += (long long)blockIdx.y * gridDim.x * K * 9;
    valid += (long long)blockIdx.y * gridDim.x * K;
    keep += (long long)blockIdx.y * gridDim.x * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    float *cx = reinterpret_cast<float *>(dsm);
    float *cy = cx + K;
    int *cid = reinterpret_cast<int *>(cy + K);
    unsigned char *sup = reinterpret_cast<unsigned char *>(cid + K);
    __shared__ int s_n, s_kept;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {                                   // ordered compaction (K <= 1024: a serial pass is fine)
        int n = 0;
        for (int i = 0; i < K; ++i)
            if (valid[(long long)b * K + i]) { cid[n] = i; ++n; }
        s_n = n;
        s_kept = 0;
    }
    for (int i = tid; i < K; i += blockDim.x) keep[(long long)b * K + i] = 0;
    __syncthreads();
    const int n = s_n;
    for (int i = tid; i < n; i += blockDim.x) {
        const float *bx = boxes + ((long long)b * K + cid[i]) * 9;
        cx[i] = bx[0]; cy[i] = bx[1]; sup[i] = 0;
    }
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (sup[i]) continue;                           // uniform: read after the barrier below
        if (s_kept >= post_max_size) break;             // keep[:post_max_size]
        const float xi = cx[i], yi = cy[i];
        for (int j = i + 1 + tid; j < n; j += blockDim.x) {
            const float dx = xi - cx[j], dy = yi - cy[j];
            if (dx * dx + dy * dy <= thresh) sup[j] = 1;
        }
        if (tid == 0) { keep[(long long)b * K + cid[i]] = 1; s_kept = s_kept + 1; }
        __syncthreads();
    }
}

// The same greedy suppression for K <= kNmsMaskK without a barrier per candidate: ordered compaction of the valid candidates
// by ballots (one candidate per thread), then ONE wave walks them in score order.  Lane l owns the candidates l, l + 64, ...
// (coordinates and a "suppressed" bit each in registers); whether candidate i is still alive is one v_readlane (i is
// wave-uniform), and only a candidate that is KEPT (<= post_max_size of them) makes the lanes test their own candidates
// against it -- O(kept * n / 64) distance tests instead of two workgroup barriers per candidate.  Same decisions in the same
// order as the plain form (`dx * dx + dy * dy <= thresh` on the same floats): ~15 us instead of ~0.5 ms per task.
constexpr int kNmsMaskK = 512;

__global__ __launch_bounds__(512) void circle_nms_kernel(int K, const float *__restrict__ boxes,
                                                         const unsigned char *__restrict__ valid, const DecodeTasks ts,
                                                         int post_max_size, unsigned char *__restrict__ keep) {
    __shared__ float cx[kNmsMaskK], cy[kNmsMaskK];
    __shared__ int cid[kNmsMaskK];
    __shared__ int wave_cnt[8];
    const float thresh = ts.nms[blockIdx.y];
    const long long row0 = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * K;          // this (task, sample)'s K candidates
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // (1) ordered compaction
    const bool v = tid < K && valid[row0 + tid] != 0;
    if (tid < K) keep[row0 + tid] = 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int pos = __popcll(m & ((1ull << lane) - 1ull)), n = 0;
    for (int w = 0; w < 8; ++w) {
        if (w < wid) pos += wave_cnt[w];
        n += wave_cnt[w];
    }
    if (v) {
        const float *bx = boxes + (row0 + tid) * 9;
        cid[pos] = tid; cx[pos] = bx[0]; cy[pos] = bx[1];
    }
    __syncthreads();
    if (wid != 0) return;
    // (2) the walk
    constexpr int R = kNmsMaskK / 64;
    float xj[R], yj[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int j = lane + 64 * t;
        xj[t] = j < n ? cx[j] : 0.f;
        yj[t] = j < n ? cy[j] : 0.f;
    }
    int rem = 0;                             // bit t: candidate lane + 64 t is suppressed
    int kept = 0;
    for (int i = 0; i < n; ++i) {            // wave-uniform
        const int r = __builtin_amdgcn_readlane(rem, i & 63);
        if ((r >> (i >> 6)) & 1) continue;
        if (kept >= post_max_size) break;    // keep[:post_max_size]
        if (lane == 0) keep[row0 + cid[i]] = 1;
        ++kept;
        const float xi = cx[i], yi = cy[i];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int j = lane + 64 * t;
            const float dx = xi - xj[t], dy = yi - yj[t];
            if (j > i && j < n && dx * dx + dy * dy <= thresh) rem |= 1 << t;
        }
    }
}

int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Tail of CenterHead.get_bboxes (mmdet3d 0.18.1; reached from layers/heads/bev_height_head.py:334-405): per sample the kept
// boxes of all tasks, task after task in candidate order, z lowered by half the height, labels offset by the classes of the
// earlier tasks.  One workgroup per sample compacts the T x K candidates in order (block scan per 256 candidates); the
// per-sample counts are the only thing the host has to read back before it can slice the outputs.
constexpr int kMergeMaxTasks = 16;
struct MergeCfg { int label_offset[kMergeMaxTasks]; };

__global__ __launch_bounds__(256) void merge_tasks_kernel(int batch, int tasks, int K, const float *__restrict__ boxes,
                                                          const float *__restrict__ scores, const int32_t *__restrict__ labels,
                                                          const unsigned char *__restrict__ keep, MergeCfg cfg,
                                                          float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                          int32_t *__restrict__ out_labels, int32_t *__restrict__ counts) {
    __shared__ int wave_sum[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    const long long total = (long long)tasks * K;
    for (int t = 0; t < tasks; ++t) {
        const long long src0 = ((long long)t * batch + b) * K;
        for (int j0 = 0; j0 < K; j0 += 256) {                      // block-uniform trip counts
            const int j = j0 + tid;
            const bool k = j < K && keep[src0 + j] != 0;
            const unsigned long long m = __ballot(k);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_sum[wid] = __popcll(m);
            __syncthreads();
            int off = base_s;
            for (int w2 = 0; w2 < wid; ++w2) off += wave_sum[w2];
            if (k) {
                const long long dst = (long long)b * total + off + before;
                const float *src = boxes + (src0 + j) * 9;
                float v[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = src[c];
                v[2] = __fsub_rn(v[2], __fmul_rn(v[5], 0.5f));     // bboxes[:, 2] - bboxes[:, 5] * 0.5, two roundings as in torch
#pragma unroll
                for (int c = 0; c < 9; ++c) out_boxes[dst * 9 + c] = v[c];
                out_scores[dst] = scores[src0 + j];
                out_labels[dst] = labels[src0 + j] + cfg.label_offset[t];
            }
            __syncthreads();
            if (tid == 0) base_s = base_s + wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
            __syncthreads();
        }
    }
    if (tid == 0) counts[b] = base_s;
}

}  // namespace

extern "C" int sgv3d_centerpoint_merge_tasks(int batch, int num_tasks, int max_num, const float *boxes, const float *scores,
                                             const int32_t *labels, const unsigned char *keep,
                                             const int32_t *classes_per_task, float *out_boxes, float *out_scores,
                                             int32_t *out_labels, int32_t *counts, void *stream) {
    SGV3D_REQUIRE(batch > 0 && num_tasks > 0 && num_tasks <= kMergeMaxTasks && max_num > 0, "centerpoint_merge_tasks: bad size");
    SGV3D_REQUIRE(boxes && scores && labels && keep && classes_per_task && out_boxes && out_scores && out_labels && counts,
                  "centerpoint_merge_tasks: null pointer");
    MergeCfg cfg;
    int off = 0;
    for (int t = 0; t < kMergeMaxTasks; ++t) {
        cfg.label_offset[t] = off;
        if (t < num_tasks) off += classes_per_task[t];
    }
    hipLaunchKernelGGL(merge_tasks_kernel, dim3(batch), dim3(256), 0, as_stream(stream), batch, num_tasks, max_num, boxes, scores,
                       labels, keep, cfg, out_boxes, out_scores, out_labels, counts);
    return check_launch("centerpoint_merge_tasks");
}

extern "C" size_t sgv3d_centerpoint_decode_workspace_bytes(int batch, int num_class, int max_num) {
    if (batch <= 0 || num_class <= 0 || max_num <= 0) return 0;
    return (sizeof(float) + sizeof(int)) * (size_t)batch * num_class * max_num + 256;
}

extern "C" size_t sgv3d_centerpoint_decode_tasks_workspace_bytes(int batch, int num_tasks, int max_class, int max_num) {
    if (batch <= 0 || num_tasks <= 0 || max_class <= 0 || max_num <= 0) return 0;
    return (sizeof(float) + sizeof(int)) * (size_t)num_tasks * batch * max_class * max_num + 256;
}

extern "C" int sgv3d_centerpoint_decode_tasks(int batch, int num_tasks, const int32_t *classes_per_task, int h, int w, int max_num,
                                              const void *const *heatmap, const void *const *reg, const void *const *height,
                                              const void *const *dim, const void *const *rot, const void *const *vel,
                                              long long batch_stride, float out_size_factor, float voxel_x, float voxel_y,
                                              float pc_x, float pc_y, float score_threshold, const float *post_center_range,
                                              int norm_bbox, const float *nms_thresh, int post_max_size, void *workspace,
                                              size_t workspace_bytes, float *boxes, float *scores, int32_t *labels,
                                              unsigned char *valid, unsigned char *keep, void *stream) {
    SGV3D_REQUIRE(batch > 0 && num_tasks > 0 && num_tasks <= kMaxTasks && h > 0 && w > 0 && max_num > 0,
                  "centerpoint_decode: non-positive size (or more than %d tasks)", kMaxTasks);
    SGV3D_REQUIRE(classes_per_task && heatmap && reg && height && dim && rot && nms_thresh && boxes && scores && labels && valid &&
                  keep && workspace, "centerpoint_decode: null pointer");
    DecodeTasks ts;
    int max_cat = 0;
    for (int t = 0; t < kMaxTasks; ++t) {
        const bool on = t < num_tasks;
        ts.heat[t] = on ? static_cast<const float *>(heatmap[t]) : nullptr;
        ts.reg[t] = on ? static_cast<const float *>(reg[t]) : nullptr;
        ts.hei[t] = on ? static_cast<const float *>(height[t]) : nullptr;
        ts.dim[t] = on ? static_cast<const float *>(dim[t]) : nullptr;
        ts.rot[t] = on ? static_cast<const float *>(rot[t]) : nullptr;
        ts.vel[t] = (on && vel) ? static_cast<const float *>(vel[t]) : nullptr;
        ts.cat[t] = on ? classes_per_task[t] : 0;
        ts.nms[t] = on ? nms_thresh[t] : 0.f;
        if (on) {
            SGV3D_REQUIRE(ts.heat[t] && ts.reg[t] && ts.hei[t] && ts.dim[t] && ts.rot[t] && ts.cat[t] > 0,
                          "centerpoint_decode: task %d has a null map or no class", t);
            max_cat = ts.cat[t] > max_cat ? ts.cat[t] : max_cat;
        }
    }
    const int Kp = pow2_ge(max_num), Kp2 = pow2_ge(max_cat * max_num);
    SGV3D_REQUIRE(Kp <= kMaxK && Kp2 <= 8192, "centerpoint_decode: max_num=%d x %d classes exceeds the LDS sort buffers", max_num, max_cat);
    SGV3D_REQUIRE((long long)h * w >= max_num, "centerpoint_decode: max_num exceeds H*W");
    const size_t need = sgv3d_centerpoint_decode_tasks_workspace_bytes(batch, num_tasks, max_cat, max_num);
    if (workspace_bytes < need) return fail(SGV3D_ENOSPACE, "centerpoint_decode: workspace has %zu bytes, needs %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float *cls_score = static_cast<float *>(workspace);
    int *cls_ind = reinterpret_cast<int *>(cls_score + (size_t)num_tasks * batch * max_cat * max_num);
    const long long hw = (long long)h * w;
    const dim3 tgrid(batch * max_cat, num_tasks);
    if (hw <= (long long)kKeyCache * kTk)
        hipLaunchKernelGGL(topk_per_class_kernel<true>, tgrid, dim3(kTk), 0, st, batch, max_cat, hw, max_num, Kp, ts, batch_stride,
                           cls_score, cls_ind);
    else
        hipLaunchKernelGGL(topk_per_class_kernel<false>, tgrid, dim3(kTk), 0, st, batch, max_cat, hw, max_num, Kp, ts, batch_stride,
                           cls_score, cls_ind);
    DecodeCfg cfg;
    cfg.out_size_factor = out_size_factor; cfg.vx = voxel_x; cfg.vy = voxel_y; cfg.pcx = pc_x; cfg.pcy = pc_y;
    cfg.score_thr = score_threshold; cfg.norm_bbox = norm_bbox; cfg.has_vel = vel != nullptr;
    cfg.has_range = post_center_range != nullptr;
    for (int i = 0; i < 6; ++i) cfg.range[i] = post_center_range ? post_center_range[i] : 0.f;
    hipLaunchKernelGGL(merge_decode_kernel, dim3(batch, num_tasks), dim3(kTk), (sizeof(float) + sizeof(int)) * (size_t)Kp2, st, batch,
                       max_cat, h, w, max_num, Kp2, cls_score, cls_ind, ts, batch_stride, cfg, boxes, scores, labels, valid);
    if (max_num <= kNmsMaskK) {
        hipLaunchKernelGGL(circle_nms_kernel, dim3(batch, num_tasks), dim3(512), 0, st, max_num, boxes, valid, ts, post_max_size, keep);
    } else {
        const size_t nms_lds = (size_t)max_num * (2 * sizeof(float) + sizeof(int) + 1) + 16;
        hipLaunchKernelGGL(circle_nms_serial_kernel, dim3(batch, num_tasks), dim3(512), nms_lds, st, max_num, boxes, valid, ts,
                           post_max_size, keep);
    }
    return check_launch("centerpoint_decode");
}

extern "C" int sgv3d_centerpoint_decode(int batch, int num_class, int h, int w, int max_num, const float *heatmap,
                                        const float *reg, const float *height, const float *dim, const float *rot,
                                        const float *vel, long long batch_stride, float out_size_factor,
                                        float voxel_x, float voxel_y, float pc_x, float pc_y, float score_threshold,
                                        const float *post_center_range, int norm_bbox, float nms_thresh,
                                        int post_max_size, void *workspace, size_t workspace_bytes, float *boxes,
                                        float *scores, int32_t *labels, unsigned char *valid, unsigned char *keep,
                                        void *stream) {
    // one task = the batched entry with a task list of one
    const int32_t cat[1] = {num_class};
    const void *hm[1] = {heatmap}, *rg[1] = {reg}, *he[1] = {height}, *dm[1] = {dim}, *ro[1] = {rot}, *ve[1] = {vel};
    const float nt[1] = {nms_thresh};
    return sgv3d_centerpoint_decode_tasks(batch, 1, cat, h, w, max_num, hm, rg, he, dm, ro, vel ? ve : nullptr, batch_stride,
                                          out_size_factor, voxel_x, voxel_y, pc_x, pc_y, score_threshold, post_center_range, norm_bbox,
                                          nt, post_max_size, workspace, workspace_bytes, boxes, scores, labels, valid, keep, stream);
}
