// Voxel pooling ("splat") for gfx950 — the operator of ops/voxel_pooling (reference:
// ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36, ops/voxel_pooling/voxel_pooling.py:10-69).
//
// Formulations, all HBM-bound indexing work (no contraction => no MFMA):
//   1. vp_atomic_kernel      reference-faithful scatter with float atomics.  One lane per (point,
//                            channel) so every atomic wave-instruction covers contiguous 256-B row
//                            segments (the shape the memory-side atomic units run at full rate);
//                            rows of dropped points are never read.
//   2. plan build + vp_gather3_kernel   deterministic CSR formulation: count -> scan -> fill ->
//                            per-segment sort once per calibration, then one launch gathers every output
//                            row with 16-B loads (work cut evenly over the sorted slot list, each voxel
//                            summed by one wave) and writes it once.  vp_gather_kernel (one wave per 4
//                            voxels) serves channel counts the slot-balanced kernel does not cover.
//   3. the same gather with rows formed on the fly as prob * context (FUSED; never materialises the
//                            [B,N,C] lifted tensor), with bf16 rows / bf16 output (bf16 compute mode), and
//                            accumulating into a caller-zeroed tensor behind the reference's own entry
//                            point (sgv3d_voxel_pooling_forward keeps a plan per stream, "level 1").
#include <atomic>
#include "common.hpp"

#include <limits.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

using namespace sgv3d;

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// shared index helper: (x, y, z) -> flat voxel id or -1   (bounds test of ..cuda.cu:24)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int voxel_of_point(const int32_t *__restrict__ geom, long long pt, int b,
                                              int X, int Y, int Z, int &x, int &y) {
    x = geom[pt * 3 + 0];
    y = geom[pt * 3 + 1];
    const int z = geom[pt * 3 + 2];
    if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) return -1;
    return (b * Y + y) * X + x;
}

// ------------------------------------------------------------------------------------------------
// 1. atomic scatter
// ------------------------------------------------------------------------------------------------
constexpr int kAtomicPts = 64;  // points per workgroup

__global__ __launch_bounds__(kBlock) void vp_atomic_kernel(
    long long total_pts, int N, int C, int X, int Y, int Z, const int32_t *__restrict__ geom,
    const float *__restrict__ feats, float *__restrict__ out, int32_t *__restrict__ pos_memo) {
    __shared__ int vid_s[kAtomicPts];
    const int tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * kAtomicPts;
    if (tid < kAtomicPts) {
        const long long pt = p0 + tid;
        int v = -1;
        if (pt < total_pts) {
            const int b = (int)(pt / N);
            int x, y;
            v = voxel_of_point(geom, pt, b, X, Y, Z, x, y);
            if (v >= 0 && pos_memo) {
                pos_memo[pt * 3 + 0] = b;
                pos_memo[pt * 3 + 1] = y;
                pos_memo[pt * 3 + 2] = x;
            }
        }
        vid_s[tid] = v;
    }
    __syncthreads();
    const int nelem = kAtomicPts * C;
    const float *f = feats + (size_t)p0 * C;
    for (int e = tid; e < nelem; e += kBlock) {
        const int pl = e / C;
        const int c = e - pl * C;
        const int v = vid_s[pl];
        if (v >= 0) {
            // no-return hardware float atomic (global_atomic_add_f32), agent scope
            __hip_atomic_fetch_add(out + (size_t)v * C + c, f[e], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2. plan (CSR) build
// ------------------------------------------------------------------------------------------------
constexpr int kScanPerThread = 8;
constexpr int kScanElems = kBlock * kScanPerThread;  // 2048 counters per workgroup

struct PlanLayout {
    long long V;        // B*Y*X
    long long total;    // B*N
    int nblk;           // scan workgroups
    int ncw;            // workgroups of the class sort
    size_t off_seg, off_cur, off_order, off_slotvox, off_blk, off_hdr, off_geom, off_long, off_perm, off_bins, bytes;
    int long_cap;       // capacity of the long-run list (entries after the count)
};

// Voxels ordered by population CLASS (vp_gather_vox_kernel, round 4): `perm` lists the voxels class after class, the largest
// populations first and the empty voxels last, IN VOXEL ORDER INSIDE A CLASS (a stable counting sort), as one 16-byte record
// {voxel id, first slot, end slot, 0} each -- a gather wave learns its work with ONE load (the kernel is bound by its chains of
// dependent loads, not by bytes), and consecutive records are spatial neighbours: their points are neighbours in the lifted
// tensor, so the rows the operator form pulls from HBM share cache lines with the rows of the wave next door (a sort by exact
// population, which the first version used, lost that: 270 against 232 us for the slot-balanced kernel on cfg-3's sparse grid).
// Classes: 0 = empty | every population 1 .. 32 its own class (the row groups of a wave then run the same number of rows: a
// coarser split -- 1-4 | 5-8 | 9-16 | 17-32 -- cost cfg-2's fused launch 2.3 us in rows added for nothing) | 33-64 | 65-96 | ... |
// 289-320 (steps of 32: the one-wave / one-workgroup border is 32 x row groups) | then steps of a factor 1.25 up to 4 655 | more
// (the workgroup-per-voxel items are dealt with a static stride: near-equal populations side by side keep the workgroups'
// shares even).  cls[c] (kVoxClasses ints after the table) = number of voxels in classes above c = position of class c's
// first record.
constexpr int kVoxClasses = 55;
__host__ __device__ inline int vp_class(int n) {
    if (n <= 0) return 0;
    if (n <= 32) return n;
    if (n <= 320) return 33 + (n - 33) / 32;            // 33 .. 41
    int c = 42, hi = 400;
    while (c < kVoxClasses - 1 && n > hi) {              // 42 .. 53: (320, 400], (400, 500], ... x 1.25
        hi += hi >> 2;
        ++c;
    }
    return c;
}

// Voxels holding more than kLongRun points ("long runs" of the sorted slot list) are listed in the plan and summed by
// dedicated workgroups of the gather launch; every other voxel is summed by the one wave whose slot range contains its
// first point (vp_gather3_kernel).  Must be >= the widest wave range (64 lanes / LPR row groups x <= 16 slots each).
constexpr int kLongRun = 128;

// Cache header of a plan (sgv3d_voxel_plan_build_cached): which geom_xyz / grid the plan was built for.
struct PlanHeader {
    int dirty;       // result of the last compare: 1 = the build kernels of this call run, 0 = they return at once
    int diff;        // accumulator of the compare workgroups
    int ticket;      // last-workgroup election of the compare kernel
    int params[7];   // magic, B, N, X, Y, Z, sort_segments of the plan held (all 0 = none)
    int builds;      // number of real builds so far (statistics / tests)
    int bar;         // vp_plan_build_one_kernel: arrivals at its grid barriers (zeroed by the compare / prologue kernel before it)
    int err;         // ... a barrier that gave up (the plan stays dirty: the level-1 entry's gated scatter serves the call)
};
constexpr int kPlanMagic = 0x53475633;  // "SGV3"

PlanLayout plan_layout(int B, int N, int X, int Y) {
    PlanLayout L;
    L.V = (long long)B * Y * X;
    L.total = (long long)B * N;
    L.nblk = cdiv(L.V, kScanElems);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.off_seg = 0;
    L.off_cur = al(L.off_seg + sizeof(int) * (size_t)(L.V + 1));
    L.off_order = al(L.off_cur + sizeof(int) * (size_t)(L.V + 1));
    L.off_slotvox = al(L.off_order + sizeof(int) * (size_t)L.total);
    L.off_blk = al(L.off_slotvox + sizeof(int) * (size_t)(L.total + 1));
    L.off_hdr = al(L.off_blk + sizeof(int) * (size_t)(L.nblk + 2));
    L.off_geom = al(L.off_hdr + sizeof(PlanHeader));
    L.off_long = al(L.off_geom + sizeof(int) * 3 * (size_t)L.total);
    L.long_cap = (int)(L.total / kLongRun) + 1;
    L.off_perm = al(L.off_long + sizeof(int) * (size_t)(L.long_cap + 1));
    L.off_bins = al(L.off_perm + sizeof(int) * 4 * (size_t)L.V);      // perm: {voxel, first slot, end slot, 0} per voxel
    L.ncw = cdiv(L.V, kBlock);                                         // workgroups of the class sort (256 voxels each)
    L.bytes = al(L.off_bins + sizeof(int) * ((size_t)kVoxClasses * L.ncw + 2 * kVoxClasses + 2));   // table | cls | class totals
    return L;
}

// Runs of equal voxel id inside one wave (consecutive frustum points along the image row mostly land
// in the same voxel): the head lane of a run issues ONE atomic for the whole run, which cuts the
// atomic count ~5-10x on real geometry and removes most of the same-address contention.
__device__ __forceinline__ void wave_runs(int v, int lane, int &head_lane, int &run_len, bool &is_head) {
    const int vp = __shfl_up(v, 1, 64);
    is_head = (lane == 0) || (vp != v);
    const unsigned long long heads = __ballot(is_head);
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    head_lane = 63 - __clzll((long long)(heads & upto));
    const unsigned long long after = head_lane == 63 ? 0ull : heads & ~((1ull << (head_lane + 1)) - 1);
    const int next = after ? __ffsll((long long)after) - 1 : 64;
    run_len = next - head_lane;
}

// Every kernel of the plan build takes `dirty` (NULL = always run): the cached build
// (sgv3d_voxel_plan_build_cached) points it at PlanHeader::dirty, written by vp_geom_compare_kernel
// earlier on the same stream, and the whole build degenerates to empty launches while geom_xyz is the
// tensor the plan was built for.  Graph-capturable: the decision is taken on the device.
#define VP_SKIP_IF_CLEAN(dirty) \
    if ((dirty) != nullptr && *reinterpret_cast<const volatile int *>(dirty) == 0) return

__global__ __launch_bounds__(kBlock) void vp_plan_init_kernel(PlanHeader *__restrict__ hdr) {
    if (threadIdx.x == 0) {
        hdr->dirty = 1; hdr->diff = 0; hdr->ticket = 0; hdr->builds = 0; hdr->bar = 0; hdr->err = 0;
        for (int i = 0; i < 7; ++i) hdr->params[i] = 0;
    }
}

// geom_xyz == the copy the plan keeps?  16-byte compares, grid-stride; the last workgroup to finish publishes the
// verdict (dirty) and re-arms the accumulators for the next call.
__global__ __launch_bounds__(kBlock) void vp_geom_compare_kernel(long long n_ints, const int32_t *__restrict__ geom,
                                                                 const int32_t *__restrict__ copy,
                                                                 PlanHeader *__restrict__ hdr, int p0, int p1, int p2,
                                                                 int p3, int p4, int p5, int p6) {
    __shared__ int any_s;
    if (threadIdx.x == 0) any_s = 0;
    __syncthreads();
    const long long n4 = n_ints >> 2;
    const long long stride = (long long)gridDim.x * kBlock;
    bool diff = false;
    const int4 *g4 = reinterpret_cast<const int4 *>(geom);
    const int4 *c4 = reinterpret_cast<const int4 *>(copy);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const int4 a = g4[i], b = c4[i];
        diff |= (a.x != b.x) | (a.y != b.y) | (a.z != b.z) | (a.w != b.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_ints & 3)) diff |= geom[n4 * 4 + threadIdx.x] != copy[n4 * 4 + threadIdx.x];
    if (diff) any_s = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (any_s) atomicOr(&hdr->diff, 1);
        __threadfence();
        const int t = atomicAdd(&hdr->ticket, 1);
        if (t == (int)gridDim.x - 1) {
            __threadfence();
            const int d = atomicOr(&hdr->diff, 0);
            const int *q = hdr->params;
            const bool same = q[0] == p0 && q[1] == p1 && q[2] == p2 && q[3] == p3 && q[4] == p4 && q[5] == p5 && q[6] == p6;
            hdr->dirty = (d != 0 || !same) ? 1 : 0;
            hdr->diff = 0;
            hdr->ticket = 0;
        }
    }
}

// closes a (re)build: the header now describes the plan held
__global__ __launch_bounds__(64) void vp_plan_commit_kernel(PlanHeader *__restrict__ hdr, int p0, int p1, int p2, int p3,
                                                            int p4, int p5, int p6) {
    if (threadIdx.x != 0 || hdr->dirty == 0) return;
    hdr->params[0] = p0; hdr->params[1] = p1; hdr->params[2] = p2; hdr->params[3] = p3;
    hdr->params[4] = p4; hdr->params[5] = p5; hdr->params[6] = p6;
    hdr->builds += 1;
    // the plan now is the one of the geom_xyz just seen: the level-1 prologue only ever RAISES dirty (no election among its
    // workgroups), so the build lowers it.  (vp_geom_compare_kernel rewrites it on every call anyway.)
    hdr->dirty = 0;
}

__device__ __forceinline__ void vp_zero_body(long long n, int *__restrict__ p, const int *__restrict__ dirty, const int VBID, const int VGRID) {
    const long long i = (long long)VBID * kBlock + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(kBlock) void vp_zero_kernel(long long n, int *__restrict__ p, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_zero_body(n, p, dirty, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void vp_count_body(long long total_pts, int N, int X, int Y, int Z,
                                                          const int32_t *__restrict__ geom,
                                                          int32_t *__restrict__ pos_memo,
                                                          int *__restrict__ count, const int *__restrict__ dirty,
                                                          int32_t *__restrict__ geom_copy, const int VBID, const int VGRID) {
    const long long pt = (long long)VBID * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int v = -1;
    if (pt < total_pts) {
        const int b = (int)(pt / N);
        int x, y;
        v = voxel_of_point(geom, pt, b, X, Y, Z, x, y);
        if (geom_copy) {        // the cached build remembers the tensor it is built for
            geom_copy[pt * 3 + 0] = x;
            geom_copy[pt * 3 + 1] = y;
            geom_copy[pt * 3 + 2] = geom[pt * 3 + 2];
        }
        if (v >= 0 && pos_memo) {
            pos_memo[pt * 3 + 0] = b;
            pos_memo[pt * 3 + 1] = y;
            pos_memo[pt * 3 + 2] = x;
        }
    }
    int head_lane, run_len;
    bool is_head;
    wave_runs(v, lane, head_lane, run_len, is_head);
    if (is_head && v >= 0) atomicAdd(count + v, run_len);
}

__global__ __launch_bounds__(kBlock) void vp_count_kernel(long long total_pts, int N, int X, int Y, int Z,
                                                          const int32_t *__restrict__ geom,
                                                          int32_t *__restrict__ pos_memo,
                                                          int *__restrict__ count, const int *__restrict__ dirty,
                                                          int32_t *__restrict__ geom_copy) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_count_body(total_pts, N, X, Y, Z, geom, pos_memo, count, dirty, geom_copy, (int)blockIdx.x, (int)gridDim.x);
}

// exclusive scan of one int per thread across a 256-thread workgroup; returns the exclusive prefix
// and leaves the workgroup total in *total_out (same value for every thread).
__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_tot /*LDS[4]*/, int &total_out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const int t = wave_tot[w];
        if (w < wid) base += t;
        tot += t;
    }
    __syncthreads();
    total_out = tot;
    return base + inc - v;
}

__device__ __forceinline__ void vp_scan_local_body(long long V, const int *__restrict__ count,
                                                               int *__restrict__ seg_start,
                                                               int *__restrict__ blk_sum, const int *__restrict__ dirty, const int VBID, const int VGRID) {
    __shared__ int wave_tot[kBlock / 64];
    const long long base = (long long)VBID * kScanElems + (long long)threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        v[i] = (base + i < V) ? count[base + i] : 0;
        sum += v[i];
    }
    int tot;
    int run = block_exclusive_scan(sum, wave_tot, tot);
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        if (base + i < V) seg_start[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) blk_sum[VBID] = tot;
}

__global__ __launch_bounds__(kBlock) void vp_scan_local_kernel(long long V, const int *__restrict__ count,
                                                               int *__restrict__ seg_start,
                                                               int *__restrict__ blk_sum, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_scan_local_body(V, count, seg_start, blk_sum, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// one workgroup: exclusive scan of blk_sum[0..nblk) in place; blk_sum[nblk] = grand total
__device__ __forceinline__ void vp_scan_top_body(int nblk, int *__restrict__ blk_sum,
                                                             const int *__restrict__ dirty, int *__restrict__ long_list, const int VBID, const int VGRID) {
    __shared__ int wave_tot[kBlock / 64];
    if (threadIdx.x == 0) long_list[0] = 0;      // re-armed for vp_long_list_kernel (runs after the scan)
    int carry = 0;
    for (int c0 = 0; c0 < nblk; c0 += kBlock) {
        const int i = c0 + threadIdx.x;
        const int v = (i < nblk) ? blk_sum[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, wave_tot, tot);
        if (i < nblk) blk_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) blk_sum[nblk] = carry;
}

__global__ __launch_bounds__(kBlock) void vp_scan_top_kernel(int nblk, int *__restrict__ blk_sum,
                                                             const int *__restrict__ dirty, int *__restrict__ long_list) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_scan_top_body(nblk, blk_sum, dirty, long_list, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void vp_scan_add_body(long long V, int nblk,
                                                             const int *__restrict__ blk_sum,
                                                             int *__restrict__ seg_start,
                                                             int *__restrict__ cursor, const int *__restrict__ dirty, const int VBID, const int VGRID) {
    const long long i = (long long)VBID * kBlock + threadIdx.x;
    if (i < V) {
        const int s = seg_start[i] + blk_sum[i / kScanElems];
        seg_start[i] = s;
        cursor[i] = s;
    } else if (i == V) {
        seg_start[V] = blk_sum[nblk];
    }
}

__global__ __launch_bounds__(kBlock) void vp_scan_add_kernel(long long V, int nblk,
                                                             const int *__restrict__ blk_sum,
                                                             int *__restrict__ seg_start,
                                                             int *__restrict__ cursor, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_scan_add_body(V, nblk, blk_sum, seg_start, cursor, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// voxels with more than kLongRun points -> long_list[1 ..], count in long_list[0] (order irrelevant: each entry is summed in
// a fixed internal order by whichever workgroup picks it up)
__device__ __forceinline__ void vp_long_list_body(long long V, const int *__restrict__ seg_start,
                                                              int *__restrict__ long_list, int cap,
                                                              const int *__restrict__ dirty, const int VBID, const int VGRID) {
    const long long v = (long long)VBID * kBlock + threadIdx.x;
    if (v >= V) return;
    if (seg_start[v + 1] - seg_start[v] > kLongRun) {
        const int i = atomicAdd(long_list, 1);
        if (i < cap) long_list[1 + i] = (int)v;
    }
}

__global__ __launch_bounds__(kBlock) void vp_long_list_kernel(long long V, const int *__restrict__ seg_start,
                                                              int *__restrict__ long_list, int cap,
                                                              const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_long_list_body(V, seg_start, long_list, cap, dirty, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void vp_fill_body(long long total_pts, int N, int X, int Y, int Z,
                                                         const int32_t *__restrict__ geom,
                                                         int *__restrict__ cursor,
                                                         int *__restrict__ order,
                                                         int *__restrict__ slot_voxel, const int *__restrict__ dirty,
                                                         const int *__restrict__ seg_start, const int VBID, const int VGRID) {
    const long long pt = (long long)VBID * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int v = -1;
    if (pt < total_pts) {
        const int b = (int)(pt / N);
        int x, y;
        v = voxel_of_point(geom, pt, b, X, Y, Z, x, y);
    }
    int head_lane, run_len;
    bool is_head;
    wave_runs(v, lane, head_lane, run_len, is_head);
    int base = 0;
    if (is_head && v >= 0) base = atomicAdd(cursor + v, run_len);
    base = __shfl(base, head_lane, 64);
    if (v >= 0) {
        const int slot = base + (lane - head_lane);
        order[slot] = (int)pt;
        // slots of a long run carry ~voxel: no gather wave owns them, the long-run workgroups sum them (vp_gather3_kernel)
        slot_voxel[slot] = (seg_start[v + 1] - seg_start[v] > kLongRun) ? ~v : v;
    }
}

__global__ __launch_bounds__(kBlock) void vp_fill_kernel(long long total_pts, int N, int X, int Y, int Z,
                                                         const int32_t *__restrict__ geom,
                                                         int *__restrict__ cursor,
                                                         int *__restrict__ order,
                                                         int *__restrict__ slot_voxel, const int *__restrict__ dirty,
                                                         const int *__restrict__ seg_start) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_fill_body(total_pts, N, X, Y, Z, geom, cursor, order, slot_voxel, dirty, seg_start, (int)blockIdx.x, (int)gridDim.x);
}

// ---- voxels by population class, stable in voxel order (perm), for the voxel-owner gather ------------------------------------
// tbl[c * ncw + b] = number of class-c voxels among the 256 voxels of workgroup b
__device__ __forceinline__ void vp_cls_count_body(long long V, const int *__restrict__ seg_start, int ncw,
                                                              int *__restrict__ tbl, const int *__restrict__ dirty, const int VBID, const int VGRID) {
    __shared__ int h[kVoxClasses];
    if (threadIdx.x < kVoxClasses) h[threadIdx.x] = 0;
    __syncthreads();
    const long long v = (long long)VBID * kBlock + threadIdx.x;
    const int c = v < V ? vp_class(seg_start[v + 1] - seg_start[v]) : -1;
#pragma unroll
    for (int k = 0; k < kVoxClasses; ++k) {
        const unsigned long long m = __ballot(c == k);
        if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(&h[k], __popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < kVoxClasses) tbl[threadIdx.x * ncw + VBID] = h[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void vp_cls_count_kernel(long long V, const int *__restrict__ seg_start, int ncw,
                                                              int *__restrict__ tbl, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_cls_count_body(V, seg_start, ncw, tbl, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// one workgroup per class: exclusive scan of the class's row of the table (workgroups ascending), in place; tot[c] = its sum
__device__ __forceinline__ void vp_cls_scan_body(int ncw, int *__restrict__ tbl, int *__restrict__ tot,
                                                             const int *__restrict__ dirty, const int VBID, const int VGRID) {
    __shared__ int wave_tot[kBlock / 64];
    const int c = VBID;
    int *row = tbl + (size_t)c * ncw;
    int carry = 0;
    for (int b0 = 0; b0 < ncw; b0 += kBlock * kScanPerThread) {
        int v[kScanPerThread], sum = 0;
        const int base = b0 + threadIdx.x * kScanPerThread;
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) {
            v[i] = base + i < ncw ? row[base + i] : 0;
            sum += v[i];
        }
        int t;
        int run = carry + block_exclusive_scan(sum, wave_tot, t);
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) {
            if (base + i < ncw) row[base + i] = run;
            run += v[i];
        }
        carry += t;
    }
    if (threadIdx.x == 0) tot[c] = carry;
}

__global__ __launch_bounds__(kBlock) void vp_cls_scan_kernel(int ncw, int *__restrict__ tbl, int *__restrict__ tot,
                                                             const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_cls_scan_body(ncw, tbl, tot, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// perm order: class kVoxClasses-1 first, class 0 (the empty voxels) last; inside a class the workgroups ascending and, inside a
// workgroup, the voxels ascending (ranks by ballots).  cls[c] = position of class c's first record (written by workgroup 0).
__device__ __forceinline__ void vp_cls_scatter_body(long long V, const int *__restrict__ seg_start, int ncw,
                                                                const int *__restrict__ tbl, const int *__restrict__ tot,
                                                                int *__restrict__ cls, int *__restrict__ perm,
                                                                const int *__restrict__ dirty, const int VBID, const int VGRID) {
    __shared__ int wcnt[kBlock / 64][kVoxClasses];
    __shared__ int cbase[kVoxClasses];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x < kVoxClasses) {
        int bsum = 0;
        for (int k = kVoxClasses - 1; k > (int)threadIdx.x; --k) bsum += tot[k];
        cbase[threadIdx.x] = bsum;
        if (VBID == 0) cls[threadIdx.x] = bsum;
    }
    const long long v = (long long)VBID * kBlock + threadIdx.x;
    int s0 = 0, s1 = 0, c = -1;
    if (v < V) {
        s0 = seg_start[v];
        s1 = seg_start[v + 1];
        c = vp_class(s1 - s0);
    }
    int rank = 0;
#pragma unroll
    for (int k = 0; k < kVoxClasses; ++k) {                       // stable: rank = earlier voxels of the same class
        const unsigned long long m = __ballot(c == k);
        if (c == k) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wid][k] = __popcll(m);
    }
    __syncthreads();
    if (c >= 0) {
        int pos = cbase[c] + tbl[c * ncw + VBID] + rank;
        for (int w = 0; w < wid; ++w) pos += wcnt[w][c];
        reinterpret_cast<int4 *>(perm)[pos] = make_int4((int)v, s0, s1, 0);
    }
}

__global__ __launch_bounds__(kBlock) void vp_cls_scatter_kernel(long long V, const int *__restrict__ seg_start, int ncw,
                                                                const int *__restrict__ tbl, const int *__restrict__ tot,
                                                                int *__restrict__ cls, int *__restrict__ perm,
                                                                const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_cls_scatter_body(V, seg_start, ncw, tbl, tot, cls, perm, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// ascending sort of order[s .. s + n), n <= 64 R, by one wave: the list lives in R registers per lane (element e = r * 64 + lane)
template <int R>
__device__ __forceinline__ void vp_sort_regs(int *__restrict__ order, int s, int n, int lane) {
    int val[R];
#pragma unroll
    for (int r = 0; r < R; ++r) val[r] = (r * 64 + lane) < n ? order[s + r * 64 + lane] : 0x7fffffff;
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int rj = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & rj) == 0) {
                        const bool up = (((r * 64 + lane) & k) == 0);
                        const int a = val[r], b = val[r | rj];
                        const int mn = min(a, b), mx = max(a, b);
                        val[r] = up ? mn : mx;
                        val[r | rj] = up ? mx : mn;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int other = __shfl_xor(val[r], j, 64);
                    const bool up = (((r * 64 + lane) & k) == 0);
                    const bool lower = (lane & j) == 0;
                    const int mn = min(val[r], other), mx = max(val[r], other);
                    val[r] = (lower == up) ? mn : mx;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if ((r * 64 + lane) < n) order[s + r * 64 + lane] = val[r];
}

// Segments of lo+1 .. 64*R points: one wave per voxel, the list lives in R registers per lane
// (element e = r*64 + lane) and is sorted by a bitonic network whose steps run over the cross-lane
// network (distance < 64) or between registers of one lane (distance >= 64): no LDS, no barrier, and
// every long voxel gets its own wave, so the per-voxel multiplicity skew costs no serialisation.
template <int R>
__device__ __forceinline__ void vp_sort_wave_body(long long V, const int *__restrict__ seg_start,
                                                              int *__restrict__ order, int lo,
                                                              const int *__restrict__ dirty, const int VBID, const int VGRID) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)VBID * (kBlock / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)VGRID * (kBlock / 64);
    // voxels are dealt to waves with a per-round rotation (97 is coprime to any power-of-two wave
    // count): long lists cluster in a few BEV columns and a plain stride would pile them on few waves
    for (long long it = 0; it * nwaves < V; ++it) {
        const long long v = it * nwaves + (wave0 + 97 * it) % nwaves;
        if (v >= V) continue;
        const int s = seg_start[v];
        const int n = seg_start[v + 1] - s;
        if (n <= lo || n > 64 * R) continue;   // wave-uniform
        vp_sort_regs<R>(order, s, n, lane);
    }
}

// The four length classes of vp_sort_wave_kernel in ONE walk over the voxels (the one-launch rebuild has few waves: a walk is
// V / waves dependent seg_start loads, and four of them were most of its time)
__device__ __forceinline__ void vp_sort_wave_all_body(long long V, const int *__restrict__ seg_start, int *__restrict__ order,
                                                      const int VBID, const int VGRID) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)VBID * (kBlock / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)VGRID * (kBlock / 64);
    for (long long it = 0; it * nwaves < V; ++it) {
        const long long v = it * nwaves + (wave0 + 97 * it) % nwaves;
        if (v >= V) continue;
        const int s = seg_start[v];
        const int n = seg_start[v + 1] - s;          // wave-uniform
        if (n <= 1 || n > 2048) continue;
        if (n <= 64) vp_sort_regs<1>(order, s, n, lane);
        else if (n <= 256) vp_sort_regs<4>(order, s, n, lane);
        else if (n <= 1024) vp_sort_regs<16>(order, s, n, lane);
        else vp_sort_regs<32>(order, s, n, lane);
    }
}

template <int R>
__global__ __launch_bounds__(kBlock) void vp_sort_wave_kernel(long long V, const int *__restrict__ seg_start,
                                                              int *__restrict__ order, int lo,
                                                              const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_sort_wave_body<R>(V, seg_start, order, lo, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// Ascending sort of every segment whose length is in (lo, hi]: normalised bitonic network (all
// comparators ascending, virtual +inf padding => works for any length), data staged in LDS when it
// fits, otherwise sorted in place in global memory by the one workgroup that owns the segment.
template <int T, int LDS_CAP>
__device__ __forceinline__ void vp_sort_segments_body(long long V, const int *__restrict__ seg_start,
                                                             int *__restrict__ order, int lo, int hi,
                                                             const int *__restrict__ dirty, const int VBID, const int VGRID) {
    __shared__ int buf[LDS_CAP];
    __shared__ int todo[T];
    __shared__ int ntodo;
    const int tid = threadIdx.x;
    // The workgroup inspects T voxels at a time and queues the ones it has to sort.  Voxel ids are
    // dealt round-robin over the workgroups (v = slot * gridDim + block): long segments cluster in
    // neighbouring voxels (near the camera), a contiguous assignment would serialise them on one CU.
    const long long G = VGRID;
    for (long long sb = 0; sb * G < V; sb += T) {
        if (tid == 0) ntodo = 0;
        __syncthreads();
        {
            // rotate the column by 37 per row: a plain v = slot*G + block would hand one BEV column
            // (all rows of one x when G == X) to one workgroup, and the hot voxels sit in few columns
            const long long v = (sb + tid) * G + (VBID + 37 * (sb + tid)) % G;
            const int n = v < V ? seg_start[v + 1] - seg_start[v] : 0;
            if (n > lo && n <= hi) todo[atomicAdd(&ntodo, 1)] = tid;   // queue order does not matter
        }
        __syncthreads();
        const int nq = ntodo;
        for (int qi = 0; qi < nq; ++qi) {
            const long long v = (sb + todo[qi]) * G + (VBID + 37 * (sb + todo[qi])) % G;
            const int s = seg_start[v];
            const int n = seg_start[v + 1] - s;
            int *data = order + s;
            const bool in_lds = n <= LDS_CAP;
            if (in_lds) {
                for (int i = tid; i < n; i += T) buf[i] = data[i];
            }
            __syncthreads();
            int *d = in_lds ? buf : data;
            int half = 1;  // npow2 / 2
            while (half * 2 < n) half <<= 1;
            for (int k = 2; (k >> 1) < n; k <<= 1) {
                const int hk = k >> 1;
                for (int i = tid; i < half; i += T) {  // flip stage: i <-> k-1-i inside each k block
                    const int blk = i / hk, off = i - blk * hk;
                    const int a = blk * k + off, b = blk * k + k - 1 - off;
                    if (b < n) {
                        const int va = d[a], vb = d[b];
                        if (va > vb) { d[a] = vb; d[b] = va; }
                    }
                }
                __syncthreads();
                for (int j = k >> 2; j > 0; j >>= 1) {  // half cleaners
                    for (int i = tid; i < half; i += T) {
                        const int a = 2 * j * (i / j) + (i % j), b = a + j;
                        if (b < n) {
                            const int va = d[a], vb = d[b];
                            if (va > vb) { d[a] = vb; d[b] = va; }
                        }
                    }
                    __syncthreads();
                }
            }
            if (in_lds) {
                for (int i = tid; i < n; i += T) data[i] = buf[i];
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

template <int T, int LDS_CAP>
__global__ __launch_bounds__(T) void vp_sort_segments_kernel(long long V, const int *__restrict__ seg_start,
                                                             int *__restrict__ order, int lo, int hi,
                                                             const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    vp_sort_segments_body<T, LDS_CAP>(V, seg_start, order, lo, hi, dirty, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// The whole (re)build as ONE launch, for the level-1 entry (sgv3d_voxel_pooling_forward[_fresh]).
//
// The build above is ~16 launches that each leave at once while geom_xyz is the plan's: harmless from a stream (the host
// enqueues them one call late, only after the device's note), but a stream CAPTURE has to record them in line -- 27 us of
// empty launches in front of a 26 us gather (59.8 us per replay at cfg-2, profiles/r04_gather_probe.txt; ROCm 7.2 has no
// conditional graph nodes, and a forked branch of the graph cost more than it hid: 75 us).  Here every phase is the body of
// its kernel above, run by kOneGrid resident workgroups over that kernel's virtual grid, with device-wide barriers between
// dependent phases:
//
//   zero | count | scan_local | scan_top | scan_add | long_list + cls_count + fill | cls_scan | cls_scatter + sorts | commit
//
// While the plan is current every workgroup leaves after one load: ONE empty launch.  The barrier is an arrival counter in the
// plan header (zeroed by the prologue kernel that precedes this one on the stream).  Progress by construction, on a plain
// (non-cooperative) launch that a stream capture can record:
//   * the grid is at most what the chip can hold of THIS kernel at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs,
//     capped at kOneGrid = 512 workgroups of 256 threads and 11 KB of LDS; vp_one_grid()), so every workgroup becomes resident
//     as soon as the finite kernels of other streams have drained from the compute units it needs -- nothing it waits for waits
//     for it;
//   * a waiting workgroup sleeps between polls and gives up after kOneSpinTicks of the 100 MHz wall clock (20 ms: five frames
//     of the cfg-2 model); giving up raises PlanHeader::err;
//   * every barrier reports the error word to its whole workgroup, and a workgroup that sees it set RETURNS: no later phase
//     runs on counts, cursors or segment bounds that an absent workgroup did not finish (they index the order / slot arrays).
//     The commit is skipped, the plan stays dirty, the entry's gated scatter serves the call and the next call tries again.
constexpr int kOneGrid = 512;
constexpr long long kOneSpinTicks = 2000000;  // wall_clock64() runs at 100 MHz: 20 ms

struct PlanOneArgs {
    long long total, V;
    int N, X, Y, Z, nblk, ncw, long_cap;
    const int32_t *geom;
    int32_t *gcopy;
    int *seg, *cur, *order, *slotvox, *blk, *long_list, *tbl, *tot, *cls, *perm;
    PlanHeader *hdr;
    int p[7];
};

// -> true (to every thread of the workgroup) when the build has failed: the caller returns.
__device__ __forceinline__ bool vp_grid_barrier(PlanHeader *hdr, int &epoch) {
    __shared__ int failed;
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        int err = __hip_atomic_load(&hdr->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err == 0) {
            __threadfence();                                                      // release this workgroup's writes
            __hip_atomic_fetch_add(&hdr->bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const int target = epoch * (int)gridDim.x;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&hdr->bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(32);            // (~1 us between polls: hundreds of workgroups poll ONE word)
                err = __hip_atomic_load(&hdr->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (err == 0 && wall_clock64() - t0 > kOneSpinTicks) {
                    __hip_atomic_store(&hdr->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    err = 1;
                }
                if (err != 0) break;
            }
            __threadfence();                                                      // acquire the others'
        }
        failed = err;
    }
    __syncthreads();
    return failed != 0;
}

__device__ __forceinline__ long long dcdiv(long long x, long long y) { return (x + y - 1) / y; }

__global__ __launch_bounds__(kBlock) void vp_plan_build_one_kernel(const PlanOneArgs a) {
    const int *dirty = &a.hdr->dirty;
    if (*reinterpret_cast<const volatile int *>(dirty) == 0) return;
    const int wg = blockIdx.x, G = gridDim.x;
    int epoch = 0;
    // a phase: the kernel's body over its virtual grid, this workgroup taking every G-th virtual block; __syncthreads between
    // virtual blocks because the bodies reuse their LDS scratch
#define VP_PHASE(NBLK, CALL)                                                    \
    for (long long vb_ = wg; vb_ < (long long)(NBLK); vb_ += G) {               \
        const int VB = (int)vb_;                                                \
        CALL;                                                                   \
        __syncthreads();                                                        \
    }
    // (bodies without LDS scratch: no barrier between virtual blocks)
#define VP_PHASE_NOSYNC(NBLK, CALL)                                             \
    for (long long vb_ = wg; vb_ < (long long)(NBLK); vb_ += G) {               \
        const int VB = (int)vb_;                                                \
        CALL;                                                                   \
    }
    const int pgrid = (int)dcdiv(a.total, kBlock);
    VP_PHASE_NOSYNC(dcdiv(a.V + 1, kBlock), vp_zero_body(a.V + 1, a.cur, nullptr, VB, 0));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE_NOSYNC(pgrid, vp_count_body(a.total, a.N, a.X, a.Y, a.Z, a.geom, nullptr, a.cur, nullptr, a.gcopy, VB, pgrid));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE(a.nblk, vp_scan_local_body(a.V, a.cur, a.seg, a.blk, nullptr, VB, a.nblk));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    if (wg == 0) vp_scan_top_body(a.nblk, a.blk, nullptr, a.long_list, 0, 1);
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE_NOSYNC(dcdiv(a.V + 1, kBlock), vp_scan_add_body(a.V, a.nblk, a.blk, a.seg, a.cur, nullptr, VB, 0));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE_NOSYNC(dcdiv(a.V, kBlock), vp_long_list_body(a.V, a.seg, a.long_list, a.long_cap, nullptr, VB, 0));
    VP_PHASE(a.ncw, vp_cls_count_body(a.V, a.seg, a.ncw, a.tbl, nullptr, VB, a.ncw));
    VP_PHASE_NOSYNC(pgrid, vp_fill_body(a.total, a.N, a.X, a.Y, a.Z, a.geom, a.cur, a.order, a.slotvox, nullptr, a.seg, VB, pgrid));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE(kVoxClasses, vp_cls_scan_body(a.ncw, a.tbl, a.tot, nullptr, VB, kVoxClasses));
    if (vp_grid_barrier(a.hdr, epoch)) return;
    VP_PHASE(a.ncw, vp_cls_scatter_body(a.V, a.seg, a.ncw, a.tbl, a.tot, a.cls, a.perm, nullptr, VB, a.ncw));
    // the five sorts touch disjoint segments (by length class): one phase; each deals the voxels over the G workgroups itself
    vp_sort_wave_all_body(a.V, a.seg, a.order, wg, G);
    __syncthreads();
    // (segments above 2048 points: LDS for up to 2048 ints here -- the stand-alone kernel stages 8192 -- longer ones are sorted in
    //  place in global memory by their workgroup; 11 KB per workgroup instead of 35)
    vp_sort_segments_body<kBlock, 2048>(a.V, a.seg, a.order, 2048, 0x7fffffff, nullptr, wg, G);
    if (vp_grid_barrier(a.hdr, epoch)) return;
#undef VP_PHASE_NOSYNC
#undef VP_PHASE
    if (wg == 0 && threadIdx.x == 0 && __hip_atomic_load(&a.hdr->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        for (int i = 0; i < 7; ++i) a.hdr->params[i] = a.p[i];
        a.hdr->builds += 1;
        __threadfence();
        a.hdr->dirty = 0;
    }
}

// Grid of vp_plan_build_one_kernel: never more workgroups than the device holds of this kernel at once (see above); SGV3D_VP_ONE_GRID
// may lower it further (diagnostics).
static int vp_one_grid() {
    static const int grid = [] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vp_plan_build_one_kernel, kBlock, 0) != hipSuccess || cus <= 0 || per_cu <= 0)
            return 64;                                   // (a grid any gfx9 part holds)
        long long fit = (long long)cus * per_cu;
        int g = (int)(fit < kOneGrid ? fit : kOneGrid);
        const char *v = getenv("SGV3D_VP_ONE_GRID");
        const int want = v ? atoi(v) : 0;
        if (want >= 32 && want < g) g = want;
        return g;
    }();
    return grid;
}


// ------------------------------------------------------------------------------------------------
// 2b/3. gather + register reduce.  One wave owns kVoxPerWave consecutive output rows; its 64 lanes
// form G = 64/LPR row groups of LPR lanes (LPR = lanes per row: C/4 float4 lanes, or C scalar lanes
// when C % 4 != 0); group g reduces points s+g, s+g+G, ... of the voxel's segment, four rows in
// flight per group, then the groups are summed through the cross-lane network in fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int kVoxPerWave = 4;
constexpr int kRowsInFlight = 4;

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vzero(float) { return 0.f; }
__device__ __forceinline__ void vacc(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vacc(float &a, const float &b) { a += b; }
__device__ __forceinline__ void vfma(float4 &a, float s, const float4 &b) {
    a.x += s * b.x; a.y += s * b.y; a.z += s * b.z; a.w += s * b.w;
}
__device__ __forceinline__ void vfma(float &a, float s, const float &b) { a += s * b; }
__device__ __forceinline__ float4 vshfl(const float4 &a, int src) {
    return make_float4(__shfl(a.x, src, 64), __shfl(a.y, src, 64), __shfl(a.z, src, 64), __shfl(a.w, src, 64));
}
__device__ __forceinline__ float vshfl(const float &a, int src) { return __shfl(a, src, 64); }

template <int VEC, bool FUSED>
__global__ __launch_bounds__(kBlock) void vp_gather_kernel(
    long long V, int C, int lpr, int groups, const int *__restrict__ seg_start,
    const int *__restrict__ order, const float *__restrict__ feats /* !FUSED: [B*N, C] */,
    const float *__restrict__ prob /* FUSED: [B*N] */, const float *__restrict__ ctx /* FUSED: [B*P, C] */,
    int N, int P, float *__restrict__ out) {
    using vec = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const bool active = g < groups;
    const int ncols = C / VEC;
    const long long v0 = wave * kVoxPerWave;
    for (int vi = 0; vi < kVoxPerWave; ++vi) {
        const long long v = v0 + vi;
        if (v >= V) return;  // wave-uniform
        const int s = __builtin_amdgcn_readfirstlane(seg_start[v]);
        const int e = __builtin_amdgcn_readfirstlane(seg_start[v + 1]);
        for (int cb = 0; cb < ncols; cb += lpr) {  // one pass unless the row is wider than a wave
            const int col = cb + cl;
            const bool col_ok = active && col < ncols;
            vec acc = vzero(vec{});
            for (int i = s + g; i < e; i += groups * kRowsInFlight) {
                int idx[kRowsInFlight];
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) {
                    const int ii = i + u * groups;
                    idx[u] = (col_ok && ii < e) ? order[ii] : -1;
                }
                vec val[kRowsInFlight];
                float pr[kRowsInFlight];
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) {
                    val[u] = vzero(vec{});
                    pr[u] = 0.f;
                    if (idx[u] >= 0) {
                        if constexpr (FUSED) {
                            const int b = idx[u] / N;
                            const int pix = (idx[u] - b * N) % P;
                            pr[u] = prob[idx[u]];
                            val[u] = *reinterpret_cast<const vec *>(ctx + ((size_t)b * P + pix) * C + (size_t)col * VEC);
                        } else {
                            val[u] = *reinterpret_cast<const vec *>(feats + (size_t)idx[u] * C + (size_t)col * VEC);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u) {
                    if constexpr (FUSED) vfma(acc, pr[u], val[u]);
                    else vacc(acc, val[u]);
                }
            }
            vec tot = acc;
            for (int gg = 1; gg < groups; ++gg) {  // fixed order: group 0 + group 1 + ...
                const vec o = vshfl(acc, lane + gg * lpr);
                vacc(tot, o);
            }
            if (g == 0 && col < ncols) *reinterpret_cast<vec *>(out + (size_t)v * C + (size_t)col * VEC) = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2c. owner-computes gather ("v3", the default for C % 4 == 0, 24 <= C <= 256).  The work is cut over the SORTED SLOTS of the
// plan, not over voxels, so every wave streams about the same number of 4C-byte rows whatever the per-voxel populations are
// (camera frustums put 10^2..10^3 points into near voxels and none into most).  A wave looks at a WINDOW of cap = groups x ch
// consecutive slots -- one chunk of ch <= 16 slots per LPR-lane row group, all rows of a chunk in flight at once -- that
// starts every W_nom = cap - margin slots, and every voxel is summed by exactly ONE wave: the one whose nominal range
// [s0, s0 + W_nom) holds the voxel's first slot.  That wave
//   * ignores the leading slots of its window that belong to a voxel begun earlier (its owner reads on into this window),
//   * runs a segmented sum per chunk and stitches the runs cut by chunk borders INSIDE the wave through the cross-lane
//     network (head / tail pieces in ascending chunk order, fixed association => deterministic),
//   * follows its last voxel past the nominal range: the margin of the window covers that in most cases with the same batch of
//     loads; beyond the window it loops (at most kLongRun slots).
// Nothing is cut between waves: no partial rows in HBM (27.6 of the 157.6 MB the round-2 gather + fix-up pair moved at cfg-2), no
// fix-up launch, no workspace, and no index lookups besides the window's own slot_voxel / order entries (where a run starts
// and ends is read off the window by ballots).  Voxels with more than kLongRun points would serialise on their owner: the plan
// marks their slots (~voxel in slot_voxel) so that no wave owns them, lists them, and the extra workgroups at the end of the
// grid sum each with all four waves.  Empty voxels are zero-filled by the same launch.
// ACC (the reference-faithful entry sgv3d_voxel_pooling_forward): rows are ADDED to what `out` holds and empty voxels are
// left alone -- the semantics of the reference's atomicAdd into a caller-zeroed tensor (voxel_pooling_forward_cuda.cu:30-33).
// ------------------------------------------------------------------------------------------------
constexpr int kChunkMax = 16;

typedef __bf16 vp_bf16x4 __attribute__((ext_vector_type(4)));

// One output row (voxel v), 4 channels per lane.  OB: the pooled map is written as bf16 with rows of ldo >= C channels, the
// padding channels zeroed (bf16 compute mode: the BEV trunk's first convolution wants a multiple of 32 input channels and
// rounds its input to bf16 anyway, so the rounding here changes no result downstream)
template <bool OB>
__device__ __forceinline__ void vp_store_row(float *out, long long v, int C, int ldo, int cl, const float4 &val) {
    if constexpr (OB) {
        __bf16 *o = reinterpret_cast<__bf16 *>(out) + (size_t)v * ldo;
        const vp_bf16x4 q = {(__bf16)val.x, (__bf16)val.y, (__bf16)val.z, (__bf16)val.w};
        *reinterpret_cast<vp_bf16x4 *>(o + (size_t)cl * 4) = q;
        if (cl < ((ldo - C) >> 2)) {
            const vp_bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            *reinterpret_cast<vp_bf16x4 *>(o + C + (size_t)cl * 4) = z;
        }
    } else {
        *reinterpret_cast<float4 *>(out + (size_t)v * C + (size_t)cl * 4) = val;
    }
}

// FB: the feature rows are bf16 (bf16 compute mode: sgv3d_lift_bf16 wrote them); sums stay f32.  FUSED: rows are formed on
// the fly as prob * context (never materialises the [B,N,C] lifted tensor).
template <bool FUSED, bool FB>
__device__ __forceinline__ float4 vp_load_row(const float *__restrict__ feats, const float *__restrict__ prob,
                                              const float *__restrict__ ctx, int idx, int cl, int C, int N, int P, float &pr) {
    if constexpr (FUSED) {
        const int b = idx / N;
        const int pix = (idx - b * N) % P;
        pr = prob[idx];
        if constexpr (FB) {                  // bf16 context rows
            const vp_bf16x4 q = *reinterpret_cast<const vp_bf16x4 *>(reinterpret_cast<const __bf16 *>(ctx) + ((size_t)b * P + pix) * C + (size_t)cl * 4);
            return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
        }
        return *reinterpret_cast<const float4 *>(ctx + ((size_t)b * P + pix) * C + (size_t)cl * 4);
    } else if constexpr (FB) {
        const vp_bf16x4 q = *reinterpret_cast<const vp_bf16x4 *>(reinterpret_cast<const __bf16 *>(feats) + (size_t)idx * C + (size_t)cl * 4);
        return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
    } else {
        return *reinterpret_cast<const float4 *>(feats + (size_t)idx * C + (size_t)cl * 4);
    }
}

template <bool OB, bool ACC>
__device__ __forceinline__ void vp_emit_row(float *out, long long v, int C, int ldo, int cl, float4 val) {
    if constexpr (ACC) {
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)v * C + (size_t)cl * 4);
        const float4 old = *o;
        val.x += old.x; val.y += old.y; val.z += old.z; val.w += old.w;
        *o = val;
    } else {
        vp_store_row<OB>(out, v, C, ldo, cl, val);
    }
}

// value of `x` held by lane `src` (any lane index per lane): one ds_bpermute per component
__device__ __forceinline__ float4 vp_from_lane(const float4 &x, int src) {
    return make_float4(__shfl(x.x, src, 64), __shfl(x.y, src, 64), __shfl(x.z, src, 64), __shfl(x.w, src, 64));
}

// Rows of the slots [base, base + groups*ch) that still belong to voxel `want` (marked or not): group g sums slots
// base + g*ch + k (k ascending) into `acc`; the caller adds the groups' sums in ascending group order.  Returns true when the
// run ended inside this batch (a slot of another voxel, or the end of the list, was seen).  Wave-uniform.
template <bool FUSED, bool FB>
__device__ __forceinline__ bool vp_sum_run_batch(float4 &acc, long long base, int T, int want, int g, int cl, int glane0, int ch,
                                                 bool ingroup, const int *__restrict__ order, const int *__restrict__ slot_voxel,
                                                 const float *__restrict__ feats, const float *__restrict__ prob,
                                                 const float *__restrict__ ctx, int C, int N, int P) {
    const long long cb = base + (long long)g * ch;
    int my_idx = -1;
    bool mine = false;
    if (ingroup && cl < ch && cb + cl < T) {
        mine = slot_voxel[cb + cl] == want;
        if (mine) my_idx = order[cb + cl];
    }
    // (slots are sorted by voxel: the slots of `want` form a prefix of the batch)
    const bool ended = __ballot(ingroup && cl < ch && !mine) != 0ull;
    float4 val[kChunkMax];
    float pr[kChunkMax];
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) {
        const int idx = __shfl(my_idx, glane0 + (k < ch ? k : 0), 64);
        val[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        pr[k] = 0.f;
        if (ingroup && k < ch && idx >= 0) val[k] = vp_load_row<FUSED, FB>(feats, prob, ctx, idx, cl, C, N, P, pr[k]);
    }
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) {
        if (k < ch) {                                   // (rows that are not `want`'s are zeros: adding them changes nothing)
            if constexpr (FUSED) vfma(acc, pr[k], val[k]);
            else vacc(acc, val[k]);
        }
    }
    return ended;
}

// A gated gather that finds its gate raised normally leaves without a store (the caller's fallback adds into a map the caller
// zeroed).  ``zero_on_gate``: the caller did NOT zero the map (sgv3d_voxel_pooling_forward_fresh) -- the gather's workgroups
// zero it on their way out, so the scatter that follows starts from zeros without a launch of its own.
__device__ __forceinline__ void vp_zero_output(void *out, unsigned long long bytes) {
    float4 *p = reinterpret_cast<float4 *>(out);
    const unsigned long long n = bytes >> 4;                   // (rows are multiples of 16 bytes)
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * kBlock)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool FUSED, bool FB = false, bool OB = false, bool ACC = false>
__global__ __launch_bounds__(kBlock) void vp_gather3_kernel(
    long long V, int C, int lpr, int groups, int ch, int w_nom, const int *__restrict__ seg_start,
    const int *__restrict__ order, const int *__restrict__ slot_voxel, const float *__restrict__ feats,
    const float *__restrict__ prob, const float *__restrict__ ctx, int N, int P, float *__restrict__ out, int ldo,
    const int *__restrict__ long_list, int long_cap, int nblk_regular,
    const int *__restrict__ gate /* NULL, or: run only while *gate == 0 */, int zero_on_gate) {
    if (gate != nullptr && *reinterpret_cast<const volatile int *>(gate) != 0) {
        if (zero_on_gate) vp_zero_output(out, (unsigned long long)V * (OB ? (unsigned long long)ldo * 2u : (unsigned long long)C * 4u));
        return;
    }
    __shared__ float4 red[kBlock / 64][64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const int glane0 = g * lpr;
    const bool ingroup = g < groups;        // (lpr == C / 4: every lane of a group owns 4 channels)
    const int src_cl = ingroup ? cl : 0;
    const int cap = groups * ch;
    const int T = seg_start[V];

    // ---------------------------------------------------------------- long runs: one workgroup per voxel at a time
    if ((int)blockIdx.x >= nblk_regular) {
        const int n_long = min(long_list[0], long_cap);
        const int stride = (int)gridDim.x - nblk_regular;
        for (int i = (int)blockIdx.x - nblk_regular; i < n_long; i += stride) {      // block-uniform trip count
            const int v = long_list[1 + i];
            const long long b = seg_start[v], e = seg_start[v + 1];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long long base = b + (long long)wid * cap; base < e; base += (long long)(kBlock / 64) * cap)
                vp_sum_run_batch<FUSED, FB>(acc, base, (int)e, ~v, g, cl, glane0, ch, ingroup, order, slot_voxel, feats, prob, ctx, C, N, P);
            float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g2 = 0; g2 < groups; ++g2) vacc(wsum, vp_from_lane(acc, g2 * lpr + src_cl));
            if (g == 0) red[wid][cl] = wsum;
            __syncthreads();
            if (wid == 0 && g == 0) {
                float4 tot = red[0][cl];
#pragma unroll
                for (int w = 1; w < kBlock / 64; ++w) vacc(tot, red[w][cl]);
                vp_emit_row<OB, ACC>(out, v, C, ldo, cl, tot);
            }
            __syncthreads();
        }
        return;
    }

    const long long wave = (long long)blockIdx.x * (kBlock / 64) + wid;
    // ---------------------------------------------------------------- rows of empty voxels (not in ACC mode)
    if constexpr (!ACC) {
        if (ingroup) {
            const long long ngroups = (long long)nblk_regular * (kBlock / 64) * groups;
            const long long per = (V + ngroups - 1) / ngroups;
            const long long v0 = (wave * groups + g) * per, v1 = min(V, v0 + per);
            for (long long v = v0; v < v1; ++v)
                if (seg_start[v] == seg_start[v + 1]) vp_store_row<OB>(out, v, C, ldo, cl, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    const long long s0 = wave * w_nom;
    if (s0 >= T) return;                                         // wave-uniform
    const long long cb = s0 + (long long)g * ch;
    // lanes 0 .. ch+1 of a group hold slot cb - 1 + cl: voxel id (all of them) and point id (the ch inner ones).  Outside the
    // list: INT_MIN, which no entry equals (entries are v >= 0, or ~v <= -1 for the slots of a long run, v < INT_MAX)
    int my_vox = INT_MIN, my_idx = -1;
    const long long my_slot = cb - 1 + cl;
    if (ingroup && cl < ch + 2 && my_slot >= 0 && my_slot < T) {
        my_vox = slot_voxel[my_slot];
        if (cl >= 1 && cl <= ch) my_idx = order[my_slot];
    }
    // where runs start inside the window: lanes cl = 1..ch of every group (slots cb .. cb+ch-1), plus the slot right behind the
    // window (cl = ch+1 of the last group).  Lane order is slot order.
    const int up_vox = __shfl_up(my_vox, 1, 64);
    const bool in_window = ingroup && cl >= 1 && (cl <= ch || (cl == ch + 1 && g == groups - 1));
    const bool starts = in_window && my_vox != up_vox;
    const unsigned long long nominal_m = __ballot(in_window && my_slot < s0 + w_nom);
    const unsigned long long start_m = __ballot(starts);
    const unsigned long long own_m = start_m & nominal_m;        // runs that start in the nominal range: this wave's
    if (own_m == 0ull) return;                                   // one voxel covers the whole range: its owner sums it
    const int first_lane = __ffsll((long long)own_m) - 1;
    const long long first = s0 + (long long)(first_lane / lpr) * ch + (first_lane % lpr) - 1;
    const unsigned long long tail_start_m = start_m & ~nominal_m;   // the first run start behind the nominal range ends the last owned run
    long long endw = s0 + cap;                                   // none inside the window: the last run goes on behind it
    if (tail_start_m != 0ull) {
        const int end_lane = __ffsll((long long)tail_start_m) - 1;
        endw = s0 + (long long)(end_lane / lpr) * ch + (end_lane % lpr) - 1;
    }
    const int prev_vox = __shfl(my_vox, glane0, 64);
    const int next_vox = __shfl(my_vox, glane0 + ch + 1, 64);
    const int lo = (int)max(0ll, min((long long)ch, first - cb));
    const int hi = (int)max(0ll, min((long long)ch, endw - cb));
    int vox[kChunkMax];
    float4 val[kChunkMax];
    float pr[kChunkMax];
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) {
        const int src = glane0 + 1 + (k < ch ? k : 0);
        const int idx = __shfl(my_idx, src, 64);
        vox[k] = __shfl(my_vox, src, 64);
        val[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        pr[k] = 0.f;
        if (ingroup && k >= lo && k < hi && vox[k] >= 0)          // (vox < 0: a long run's slot, or behind the end of the list)
            val[k] = vp_load_row<FUSED, FB>(feats, prob, ctx, idx, cl, C, N, P, pr[k]);
    }
    // ---------------------------------------------------------------- segmented sum of the chunk
    float4 head = make_float4(0.f, 0.f, 0.f, 0.f), tail = head, acc = head;
    bool has_head = false, single = false, has_tail = false;
    int tail_vox = -1;
    if (ingroup && lo < hi) {
        int cur = vox[0];
        bool first_run = true;
        // the first run began in an earlier chunk of this wave (and is one this wave sums)
        const bool before = lo == 0 && cb > first && prev_vox == vox[0] && vox[0] >= 0;
#pragma unroll
        for (int k = 0; k < kChunkMax; ++k) {
            if (k >= lo && k < hi) {
                if (k == lo) cur = vox[k];
                if (vox[k] != cur) {                                     // the run of `cur` ended inside the chunk
                    if (cur >= 0) {
                        if (first_run && before) { head = acc; has_head = true; }
                        else vp_emit_row<OB, ACC>(out, cur, C, ldo, cl, acc);
                    }
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    cur = vox[k];
                    first_run = false;
                }
                if constexpr (FUSED) vfma(acc, pr[k], val[k]);
                else vacc(acc, val[k]);
            }
        }
        if (cur >= 0) {
            const bool after = hi == ch && next_vox == cur;                  // continues in the next chunk / behind the window
            if (first_run && before) { head = acc; has_head = true; single = after; }
            else if (after) { tail = acc; has_tail = true; tail_vox = cur; }
            else vp_emit_row<OB, ACC>(out, cur, C, ldo, cl, acc);
        }
    }
    // ---------------------------------------------------------------- stitch the runs cut by chunk borders (ascending chunks)
    const unsigned long long head_m = __ballot(has_head && cl == 0);
    const unsigned long long single_m = __ballot(single && cl == 0);
    const unsigned long long tail_m = __ballot(has_tail && cl == 0);
    if ((head_m | tail_m) == 0ull) return;                        // wave-uniform
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    int carry_vox = -1;
    for (int g2 = 0; g2 < groups; ++g2) {
        const int l0 = g2 * lpr;
        if ((head_m >> l0) & 1ull) {
            vacc(carry, vp_from_lane(head, l0 + src_cl));
            if (!((single_m >> l0) & 1ull)) {                     // the run ends in this chunk: its row is complete
                if (g == 0) vp_emit_row<OB, ACC>(out, carry_vox, C, ldo, cl, carry);
                carry_vox = -1;
            }
        }
        if ((tail_m >> l0) & 1ull) {
            carry = vp_from_lane(tail, l0 + src_cl);
            carry_vox = __builtin_amdgcn_readfirstlane(__shfl(tail_vox, l0, 64));
        }
    }
    if (carry_vox < 0) return;
    // ---------------------------------------------------------------- the last run goes on behind the window (<= kLongRun slots)
    float4 eacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long base = s0 + cap; base < T; base += cap)
        if (vp_sum_run_batch<FUSED, FB>(eacc, base, T, carry_vox, g, cl, glane0, ch, ingroup, order, slot_voxel, feats, prob, ctx, C, N, P))
            break;
    for (int g2 = 0; g2 < groups; ++g2) vacc(carry, vp_from_lane(eacc, g2 * lpr + src_cl));
    if (g == 0) vp_emit_row<OB, ACC>(out, carry_vox, C, ldo, cl, carry);
}

// ------------------------------------------------------------------------------------------------
// 2d. the same owner-computes gather, written for instruction count.  The generic kernel above runs ~1700 wave-instructions
// per wave; at 11.4 waves per SIMD and 4 issue cycles per vector instruction that is 77 k cycles = 38 us on its own -- the
// launch was issue-bound at 4.0 TB/s, not memory-bound.  Here (non-FUSED, tensors below 4 GB):
//   * every row load / row store is a raw buffer instruction with a 32-bit offset idx * row_bytes + lane_bytes, and a dead
//     slot (not this wave's, a long run's, outside the list) carries idx = -1 / voxel = -1, whose offset lies beyond
//     num_records: the load returns zeros and the store is dropped by the hardware -- no predication, no 64-bit address math;
//   * liveness is decided once, by the lanes that fetched the window's slot_voxel / order entries, and handed to the row
//     lanes through LDS (9 ds_read_b128 instead of 34 ds_bpermute);
//   * the segmented sum takes one wave-uniform branch per slot (any row group at a run border?) with a branch-free body.
// The rare paths (last run longer than the window, long-run workgroups) are out of line.
// ------------------------------------------------------------------------------------------------
typedef int vp_i32x4 __attribute__((ext_vector_type(4)));
typedef float vp_f32x4 __attribute__((ext_vector_type(4)));
typedef float vp_f32x2 __attribute__((ext_vector_type(2)));

struct VpFastArgs {
    const int *seg_start, *order, *slot_voxel, *long_list;
    const void *feats;
    void *out;
    const int *gate;
    unsigned feat_bytes, out_bytes;   // num_records of the two buffers
    int V, C, lpr, groups, ch, w_nom, ldo, long_cap, nblk_regular, N;
    // FUSED (lift-splat): feats = context [B, P, C] f32, rows formed on the fly as prob[point] * context[pixel of the point]
    const float *prob;
    int P;
    int zero_on_gate;       // gate raised: zero the output instead of leaving it alone (vp_zero_output)
};

// FUSED: a live slot's point id becomes the row of its pixel in the context tensor, `pr` its probability (0 for dead slots).
// Done once per slot by the lane that fetched the slot's entry (two integer divisions per window, not per row).
template <bool FUSED, class Args>
__device__ __forceinline__ void vp_fused_index(const Args &a, int &idx, float &pr) {
    if constexpr (FUSED) {
        pr = 0.f;
        if (idx >= 0) {
            pr = a.prob[idx];
            const int b = (int)((unsigned)idx / (unsigned)a.N);
            const int rem = idx - b * a.N;
            idx = b * a.P + (int)((unsigned)rem % (unsigned)a.P);
        }
    }
}

// acc += row (FUSED: the row is pr * context row, the product rounded to f32 first -- the bits of the materialised lifted tensor)
template <bool FUSED>
__device__ __forceinline__ void vp_add_row(float4 &acc, const float4 &v, float pr) {
    if constexpr (FUSED) {
#pragma clang fp contract(off)      // (a fused multiply-add would skip the product's rounding)
        const float px = pr * v.x, py = pr * v.y, pz = pr * v.z, pw = pr * v.w;
        acc.x += px; acc.y += py; acc.z += pz; acc.w += pw;
    } else {
        vacc(acc, v);
    }
}

template <bool FB>
__device__ __forceinline__ float4 vp_buf_load_row(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    if constexpr (FB) {
        const vp_f32x2 raw = __builtin_bit_cast(vp_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
        const vp_bf16x4 q = __builtin_bit_cast(vp_bf16x4, raw);
        return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
    } else {
        const vp_f32x4 v = __builtin_bit_cast(vp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}

// row of voxel `vox` (-1: dropped by the hardware).  lane_out: byte offset of the lane inside a row.
template <bool OB, bool ACC>
__device__ __forceinline__ void vp_buf_emit(__amdgpu_buffer_rsrc_t rsrc, int vox, unsigned row_bytes, unsigned lane_out,
                                            unsigned pad_off, float4 v) {
    const unsigned off = (unsigned)vox * row_bytes + lane_out;
    if constexpr (OB) {
        const vp_bf16x4 q = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(vp_f32x2, q), rsrc, off, 0, 0);
        // zeroed padding channels (pad_off = ~0u marks the lanes that own none: their store goes to "voxel -1" and is dropped)
        const vp_bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        const unsigned poff = pad_off == ~0u ? ~0u - 8u : (unsigned)vox * row_bytes + pad_off;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(vp_f32x2, z), rsrc, poff, 0, 0);
    } else {
        if constexpr (ACC) {
            const vp_f32x4 old = __builtin_bit_cast(vp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            v.x += old[0]; v.y += old[1]; v.z += old[2]; v.w += old[3];
        }
        const vp_f32x4 o = {v.x, v.y, v.z, v.w};
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, off, 0, 0);
    }
}

// Rows of the slots [base, base + groups*ch) that still belong to voxel `want` (marked or not), lean form: group g sums slots
// base + g*ch + k (k ascending) into `acc`; the caller adds the groups' sums in ascending group order.  Returns true when
// the run ended inside this batch.  `idw`: the wave's LDS index area.  Wave-uniform.
template <bool FB, bool FUSED>
__device__ __forceinline__ bool vp_fast_run_batch(float4 &acc, int base, int T, int want, int g, int cl, int ch, bool ingroup,
                                                  int gs, const int *__restrict__ order, const int *__restrict__ slot_voxel,
                                                  __amdgpu_buffer_rsrc_t f_rsrc, unsigned row_in, unsigned lane_in, int *idw,
                                                  const VpFastArgs &a, float *prw) {
    const int slot = base + g * ch + cl;
    const bool index_lane = ingroup && cl < ch;
    bool mine = false;
    int my_idx = -1;
    if (index_lane && slot < T) {
        mine = slot_voxel[slot] == want;
        if (mine) my_idx = order[slot];
    }
    const bool ended = __ballot(index_lane && !mine) != 0ull;     // (sorted by voxel: `want`'s slots are a prefix of the batch)
    float my_pr = 0.f;
    vp_fused_index<FUSED>(a, my_idx, my_pr);
    if (index_lane) {
        idw[g * kChunkMax + cl] = my_idx;
        if constexpr (FUSED) prw[g * kChunkMax + cl] = my_pr;
    }
    const vp_i32x4 *ip = reinterpret_cast<const vp_i32x4 *>(idw + gs * kChunkMax);
    int idx[kChunkMax];
#pragma unroll
    for (int q = 0; q < kChunkMax / 4; ++q) {
        const vp_i32x4 t = ip[q];
        idx[4 * q + 0] = t[0]; idx[4 * q + 1] = t[1]; idx[4 * q + 2] = t[2]; idx[4 * q + 3] = t[3];
    }
    float pr[kChunkMax];
    if constexpr (FUSED) {
        const vp_f32x4 *pp = reinterpret_cast<const vp_f32x4 *>(prw + gs * kChunkMax);
#pragma unroll
        for (int q = 0; q < kChunkMax / 4; ++q) {
            const vp_f32x4 t = pp[q];
            pr[4 * q + 0] = t[0]; pr[4 * q + 1] = t[1]; pr[4 * q + 2] = t[2]; pr[4 * q + 3] = t[3];
        }
    }
    float4 val[kChunkMax];
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) val[k] = vp_buf_load_row<FB>(f_rsrc, (unsigned)idx[k] * row_in + lane_in);
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k)
        if (k < ch) vp_add_row<FUSED>(acc, val[k], FUSED ? pr[k] : 0.f);   // (rows that are not `want`'s came back as zeros)
    return ended;
}

constexpr int kEvStride = 20;    // ints per row group in the LDS image of the window: entries cl = 0 .. ch+1 (<= 18), 16-B rows
constexpr int kMaxGroups = 10;   // 64 lanes / LPR >= 6

template <bool FB, bool OB, bool ACC, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void vp_gather_fast_kernel(const VpFastArgs a) {
    if (a.gate != nullptr && *reinterpret_cast<const volatile int *>(a.gate) != 0) {
        if (a.zero_on_gate) vp_zero_output(a.out, a.out_bytes);
        return;
    }
    // per wave: ev[group][cl] = voxel of slot cb-1+cl if this wave sums it, else -1; idx[group][k] = its point id, else -1;
    // one more all-dead block for the lanes that belong to no row group
    __shared__ __attribute__((aligned(16))) int ev_s[kBlock / 64][(kMaxGroups + 1) * kEvStride];
    __shared__ __attribute__((aligned(16))) int idx_s[kBlock / 64][(kMaxGroups + 1) * kChunkMax];
    __shared__ float4 red[kBlock / 64][64];
    __shared__ __attribute__((aligned(16))) float pr_s[kBlock / 64][FUSED ? (kMaxGroups + 1) * kChunkMax : 4];   // FUSED: idx's probabilities
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int lpr = a.lpr, groups = a.groups, ch = a.ch;
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const bool ingroup = g < groups;
    const int gs = ingroup ? g : groups;                         // LDS block of the lane: its row group, or the all-dead one
    const int src_cl = ingroup ? cl : 0;
    const int cap = groups * ch;
    const int T = a.seg_start[a.V];
    int *evw = ev_s[wid], *idw = idx_s[wid];
    float *prw = pr_s[wid];
    if (lane < kChunkMax) {                                      // the all-dead index block
        idw[groups * kChunkMax + lane] = -1;
        if constexpr (FUSED) prw[groups * kChunkMax + lane] = 0.f;
    }
    const unsigned row_in = (unsigned)a.C * (FB ? 2u : 4u);
    const unsigned row_out = OB ? (unsigned)a.ldo * 2u : (unsigned)a.C * 4u;
    const unsigned lane_in = (unsigned)cl * (FB ? 8u : 16u);
    const unsigned lane_out = (unsigned)cl * (OB ? 8u : 16u);
    unsigned pad_off = ~0u;
    if constexpr (OB) {
        if (cl < ((a.ldo - a.C) >> 2)) pad_off = (unsigned)a.C * 2u + (unsigned)cl * 8u;
    }
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.feats), 0, (int)a.feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    // ---------------------------------------------------------------- long runs: one workgroup per voxel at a time
    if ((int)blockIdx.x >= a.nblk_regular) {
        const int n_long = min(a.long_list[0], a.long_cap);
        const int stride = (int)gridDim.x - a.nblk_regular;
        for (int i = (int)blockIdx.x - a.nblk_regular; i < n_long; i += stride) {      // block-uniform trip count
            const int v = a.long_list[1 + i];
            const int b = a.seg_start[v], e = a.seg_start[v + 1];
            float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int base = b + wid * cap; base < e; base += (kBlock / 64) * cap)
                vp_fast_run_batch<FB, FUSED>(racc, base, e, ~v, g, cl, ch, ingroup, gs, a.order, a.slot_voxel, f_rsrc, row_in, lane_in, idw, a, prw);
            float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g2 = 0; g2 < groups; ++g2) vacc(wsum, vp_from_lane(racc, g2 * lpr + src_cl));
            if (g == 0) red[wid][cl] = wsum;
            __syncthreads();
            if (wid == 0) {
                float4 tot = red[0][src_cl];
#pragma unroll
                for (int w = 1; w < kBlock / 64; ++w) vacc(tot, red[w][src_cl]);
                vp_buf_emit<OB, ACC>(o_rsrc, g == 0 ? v : -1, row_out, lane_out, pad_off, tot);
            }
            __syncthreads();
        }
        return;
    }
    const int wave = (int)blockIdx.x * (kBlock / 64) + wid;

    // ---------------------------------------------------------------- rows of empty voxels (not in ACC mode)
    if constexpr (!ACC) {
        const int ngroups = a.nblk_regular * (kBlock / 64) * groups;
        const int per = (a.V + ngroups - 1) / ngroups;
        const int v0 = (wave * groups + (ingroup ? g : 0)) * per;
        for (int i = 0; i < per; ++i) {                           // uniform trip count
            const int v = v0 + i;
            int vz = -1;
            if (ingroup && v < a.V && a.seg_start[v] == a.seg_start[v + 1]) vz = v;
            vp_buf_emit<OB, false>(o_rsrc, vz, row_out, lane_out, pad_off, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    const int s0 = wave * a.w_nom;
    if (s0 >= T) return;                                         // wave-uniform
    const int cb = s0 + g * ch;
    // ---------------------------------------------------------------- the window's entries: lane cl of a group holds slot cb-1+cl
    const int my_slot = cb - 1 + cl;
    int my_vox = INT_MIN, my_idx = -1;
    if (ingroup && cl < ch + 2 && my_slot >= 0 && my_slot < T) {
        my_vox = a.slot_voxel[my_slot];
        my_idx = a.order[my_slot];
    }
    const int up_vox = __shfl_up(my_vox, 1, 64);
    const bool in_window = ingroup && cl >= 1 && (cl <= ch || (cl == ch + 1 && g == groups - 1));
    const unsigned long long start_m = __ballot(in_window && my_vox != up_vox);
    const unsigned long long nominal_m = __ballot(in_window && my_slot < s0 + a.w_nom);
    const unsigned long long own_m = start_m & nominal_m;
    if (own_m == 0ull) return;                                   // one voxel covers the whole range: its owner sums it
    const int first_lane = __ffsll((long long)own_m) - 1;
    const int first = s0 + (first_lane / lpr) * ch + (first_lane % lpr) - 1;
    const unsigned long long tail_start_m = start_m & ~nominal_m;
    int endw = s0 + cap;
    if (tail_start_m != 0ull) {
        const int end_lane = __ffsll((long long)tail_start_m) - 1;
        endw = s0 + (end_lane / lpr) * ch + (end_lane % lpr) - 1;
    }
    // (slots outside the list hold INT_MIN.)  The slot right behind the window counts as live when no run starts in the tail of
    // the window: the chunk before it then sees its last run continue and hands it to the extension loop below
    const bool live = my_vox >= 0 && my_slot >= first && (my_slot < endw || (tail_start_m == 0ull && my_slot == s0 + cap));
    int row_idx = live ? my_idx : -1;
    float my_pr = 0.f;
    vp_fused_index<FUSED>(a, row_idx, my_pr);
    if (ingroup && cl < ch + 2) {
        evw[g * kEvStride + cl] = live ? my_vox : -1;
        if (cl >= 1 && cl <= ch) {
            idw[g * kChunkMax + cl - 1] = row_idx;
            if constexpr (FUSED) prw[g * kChunkMax + cl - 1] = my_pr;
        }
    }
    if (lane < kEvStride) evw[groups * kEvStride + lane] = -1;   // the all-dead block
    // ---------------------------------------------------------------- all rows of the chunk in flight
    float4 val[kChunkMax];
    float pr[kChunkMax];
    if constexpr (FUSED) {
        const vp_f32x4 *pp = reinterpret_cast<const vp_f32x4 *>(prw + gs * kChunkMax);
#pragma unroll
        for (int q = 0; q < kChunkMax / 4; ++q) {
            const vp_f32x4 t = pp[q];
            pr[4 * q + 0] = t[0]; pr[4 * q + 1] = t[1]; pr[4 * q + 2] = t[2]; pr[4 * q + 3] = t[3];
        }
    }
    {
        const vp_i32x4 *ip = reinterpret_cast<const vp_i32x4 *>(idw + gs * kChunkMax);
        int idx[kChunkMax];
#pragma unroll
        for (int q = 0; q < kChunkMax / 4; ++q) {
            const vp_i32x4 t = ip[q];
            idx[4 * q + 0] = t[0]; idx[4 * q + 1] = t[1]; idx[4 * q + 2] = t[2]; idx[4 * q + 3] = t[3];
        }
#pragma unroll
        for (int k = 0; k < kChunkMax; ++k)                       // (entries k >= ch are stale: never added below)
            val[k] = vp_buf_load_row<FB>(f_rsrc, (unsigned)idx[k] * row_in + lane_in);
    }
    int ev[kEvStride];
    {
        const vp_i32x4 *ep = reinterpret_cast<const vp_i32x4 *>(evw + gs * kEvStride);
#pragma unroll
        for (int q = 0; q < kEvStride / 4; ++q) {
            const vp_i32x4 t = ep[q];
            ev[4 * q + 0] = t[0]; ev[4 * q + 1] = t[1]; ev[4 * q + 2] = t[2]; ev[4 * q + 3] = t[3];
        }
    }
    // ---------------------------------------------------------------- segmented sum of the chunk: ev[1 + k] is slot k's voxel
    float4 head = make_float4(0.f, 0.f, 0.f, 0.f), acc = head;
    const bool before = ev[0] == ev[1] && ev[1] >= 0;            // the first run began in the previous chunk of this wave
    bool first_run = true, has_head = false;
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) {
        if (k < ch) {                                            // uniform
            if (k > 0) {
                const bool bnd = ev[k + 1] != ev[k];             // the run of slot k-1 ended
                if (__ballot(bnd) != 0ull) {                     // wave-uniform branch, branch-free body
                    const bool to_head = bnd && first_run && before;
                    head.x = to_head ? acc.x : head.x; head.y = to_head ? acc.y : head.y;
                    head.z = to_head ? acc.z : head.z; head.w = to_head ? acc.w : head.w;
                    has_head = has_head || to_head;
                    vp_buf_emit<OB, ACC>(o_rsrc, (bnd && !to_head) ? ev[k] : -1, row_out, lane_out, pad_off, acc);
                    acc.x = bnd ? 0.f : acc.x; acc.y = bnd ? 0.f : acc.y; acc.z = bnd ? 0.f : acc.z; acc.w = bnd ? 0.f : acc.w;
                    first_run = first_run && !bnd;
                }
            }
            vp_add_row<FUSED>(acc, val[k], FUSED ? pr[k] : 0.f);
        }
    }
    // the chunk's last run: slot ch-1 is ev[ch]; the slot behind the chunk ev[ch + 1] (ch is uniform but not a constant:
    // read the two entries back from LDS instead of indexing the register array dynamically)
    const int last_vox = evw[gs * kEvStride + ch];
    const int next_vox = evw[gs * kEvStride + ch + 1];
    const bool after = next_vox == last_vox && last_vox >= 0;    // continues in the next chunk / behind the window
    const bool last_is_head = first_run && before;
    bool single = false, has_tail = false;
    float4 tail = make_float4(0.f, 0.f, 0.f, 0.f);
    if (last_is_head) { head = acc; has_head = true; single = after; }
    else if (after) { tail = acc; has_tail = true; }
    vp_buf_emit<OB, ACC>(o_rsrc, (!last_is_head && !after) ? last_vox : -1, row_out, lane_out, pad_off, acc);
    // ---------------------------------------------------------------- stitch the runs cut by chunk borders (ascending chunks)
    const unsigned long long head_m = __ballot(ingroup && has_head && cl == 0);
    const unsigned long long single_m = __ballot(ingroup && single && cl == 0);
    const unsigned long long tail_m = __ballot(ingroup && has_tail && cl == 0);
    if ((head_m | tail_m) == 0ull) return;                        // wave-uniform
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    int carry_vox = -1;
    for (int g2 = 0; g2 < groups; ++g2) {
        const int l0 = g2 * lpr;
        if ((head_m >> l0) & 1ull) {
            vacc(carry, vp_from_lane(head, l0 + src_cl));
            if (!((single_m >> l0) & 1ull)) {                     // the run ends in this chunk: its row is complete
                vp_buf_emit<OB, ACC>(o_rsrc, g == 0 ? carry_vox : -1, row_out, lane_out, pad_off, carry);
                carry_vox = -1;
            }
        }
        if ((tail_m >> l0) & 1ull) {
            carry = vp_from_lane(tail, l0 + src_cl);
            carry_vox = evw[g2 * kEvStride + ch];                 // (uniform address: that chunk's last voxel)
        }
    }
    if (carry_vox < 0) return;
    // ---------------------------------------------------------------- the last run goes on behind the window (<= kLongRun slots)
    float4 eacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = s0 + cap; base < T; base += cap)
        if (vp_fast_run_batch<FB, FUSED>(eacc, base, T, carry_vox, g, cl, ch, ingroup, gs, a.order, a.slot_voxel, f_rsrc, row_in, lane_in, idw, a, prw))
            break;
    for (int g2 = 0; g2 < groups; ++g2) vacc(carry, vp_from_lane(eacc, g2 * lpr + src_cl));
    vp_buf_emit<OB, ACC>(o_rsrc, g == 0 ? carry_vox : -1, row_out, lane_out, pad_off, carry);
}

// ------------------------------------------------------------------------------------------------
// 2e. voxel-owner gather (round 4).  The slot-balanced kernel above spends ~20 wave-instructions per slot on the
// book-keeping of runs that start and end anywhere inside a wave's window (ballots, per-slot boundary selects, stitching);
// it is issue-bound -- harmless for the operator, whose rows come from HBM, but the fused lift-splat form reads its rows from
// the L2-resident context map and ran at 0.15 of its byte roofline.  Here NOTHING is cut at arbitrary places: the plan lists
// the voxels by population class (perm, cls) and a row group of lanes owns a whole voxel (or a whole, evenly cut
// piece of one), so the inner loop is "16 rows in flight, 16 adds" and a voxel is emitted exactly once:
//   * population <= 32                one row group per voxel, `groups` voxels per wave (neighbours in perm are of one
//                                     population class, so the groups of a wave finish together); <= 4 / <= 8 points: four /
//                                     two voxels share a row group's batch;
//   * 32 < population <= 32 * groups  one wave per voxel: `groups` equal pieces, added in piece order;
//   * larger                          one workgroup per voxel: 4 * groups equal pieces; a wave adds its pieces in order, the
//                                     four wave sums meet in LDS and are added in wave order.
// Fixed association per (population, channel count) => bit-reproducible.  Work is dealt with a static stride over a grid that
// is resident at once (perm goes from the largest class to the smallest, so every wave gets the same mix and the largest
// voxels start first); rows of empty voxels are zeroed by a walk over contiguous voxel ranges.  All loops are bounded by
// counts read from the plan, every wave reaches the end of the kernel.
// ------------------------------------------------------------------------------------------------
struct VpVoxArgs {
    const int *order, *cls, *seg_start;     // cls[c] = number of voxels in population classes above c (vp_class)
    const int4 *perm;       // {voxel, first slot, end slot, 0} by descending population
    const void *feats;
    void *out;
    const int *gate;
    unsigned feat_bytes, out_bytes;
    int V, C, lpr, groups, vb, ldo, N;
    const float *prob;      // FUSED (lift-splat): feats = context [B, P, C], rows formed as prob[point] * context[pixel]
    int P;
    unsigned magN, magP;    // exact division of a point id (< 2^31) by N / P: umulhi(id, mag) >> sh (vp_magic)
    int shN, shP;
    int dbg;                // SGV3D_VP_DEBUG probe bits (launch_gather)
    int zero_on_gate;       // gate raised: zero the output instead of leaving it alone (vp_zero_output)
};

// floor(n / d) for 0 <= n < 2^31, d >= 2, as umulhi(n, mag) >> sh (a round-up magic number of 32 bits is exact for 31-bit
// dividends); the fused gather turns every point id into (sample, pixel) with it instead of two hardware-emulated divisions
// max over the row groups of a group-uniform value (lane g * lpr speaks for group g): `groups` v_readlane, no LDS crossbar
// (six dependent ds_bpermute of a butterfly cost ~400 cycles of latency per work item)
__device__ __forceinline__ int vp_group_max(int v, int lpr, int groups) {
    int m = __builtin_amdgcn_readlane(v, 0);
    for (int g2 = 1; g2 < groups; ++g2) m = max(m, __builtin_amdgcn_readlane(v, g2 * lpr));
    return m;
}
struct VpMagic { unsigned mag; int sh; };
VpMagic vp_magic(int d) {
    VpMagic m{0u, 0};
    if (d < 2) return m;                                        // (d == 1 is handled by the caller's select)
    int l = 0;
    while ((1ll << l) < d) ++l;
    m.mag = (unsigned)(((1ull << (31 + l)) / (unsigned long long)d) + 1ull);
    m.sh = l - 1;
    return m;
}
__device__ __forceinline__ int vp_fast_div(int n, int d, unsigned mag, int sh) {
    return d == 1 ? n : (int)(__umulhi((unsigned)n, mag) >> sh);
}
constexpr int kVoxGrid = 1024;   // workgroups of the smallest launch (4 per CU); larger problems take 2 x / 4 x (launch_gather)

// acc += rows of the slots [pb, pe) in slot order, `vb` <= VB at a time; `maxlen` >= pe - pb is wave-uniform (the longest piece
// of the wave), slots past a group's own end are dead (index -1: the load returns zeros).  What bounds this loop is its
// chain of dependent loads (slot -> point id -> row): the probabilities of the fused form are requested BEFORE the rows and
// waited for after the rows are in flight, so they add no round trip.
template <bool FB, bool FUSED, int VB>
__device__ __forceinline__ void vp_vox_piece(float4 &acc, int pb, int pe, int maxlen, int cl, bool ingroup, int g, int gs, int vb,
                                             __amdgpu_buffer_rsrc_t f_rsrc, unsigned row_in, unsigned lane_in, int *idw,
                                             float *prw, const VpVoxArgs &a) {
    const bool index_lane = ingroup && cl < vb;
    for (int base = 0; base < maxlen; base += vb) {              // wave-uniform trip count
        const int slot = pb + base + cl;
        int my_idx = -1;
        if (index_lane && slot < pe) my_idx = a.order[slot];
        float my_pr = 0.f;
        int my_row = my_idx;
        if constexpr (FUSED) {
            if (my_idx >= 0) {
                my_pr = a.prob[my_idx];                          // (in flight under the row loads below)
                const int b = vp_fast_div(my_idx, a.N, a.magN, a.shN);
                const int rem = my_idx - b * a.N;
                my_row = b * a.P + rem - vp_fast_div(rem, a.P, a.magP, a.shP) * a.P;
            }
        }
        if (index_lane) idw[g * VB + cl] = my_row;
        const vp_i32x4 *ip = reinterpret_cast<const vp_i32x4 *>(idw + gs * VB);
        int idx[VB];
#pragma unroll
        for (int q = 0; q < VB / 4; ++q) {
            const vp_i32x4 t = ip[q];
            idx[4 * q + 0] = t[0]; idx[4 * q + 1] = t[1]; idx[4 * q + 2] = t[2]; idx[4 * q + 3] = t[3];
        }
        float4 val[VB];
#pragma unroll
        for (int k = 0; k < VB; ++k) val[k] = vp_buf_load_row<FB>(f_rsrc, (a.dbg & 4) ? ~0u - 64u : (unsigned)idx[k] * row_in + lane_in);
        float pr[VB];
        if constexpr (FUSED) {
            if (index_lane) prw[g * VB + cl] = my_pr;
            const vp_f32x4 *pp = reinterpret_cast<const vp_f32x4 *>(prw + gs * VB);
#pragma unroll
            for (int q = 0; q < VB / 4; ++q) {
                const vp_f32x4 t = pp[q];
                pr[4 * q + 0] = t[0]; pr[4 * q + 1] = t[1]; pr[4 * q + 2] = t[2]; pr[4 * q + 3] = t[3];
            }
        }
        const int n = min(vb, maxlen - base);                    // (entries k >= vb stay dead: never written, never added)
#pragma unroll
        for (int k = 0; k < VB; ++k)
            if (k < n) vp_add_row<FUSED>(acc, val[k], FUSED ? pr[k] : 0.f);
    }
}

// J small voxels per row group in ONE batch of VB slots: voxel j of the group owns the slots [j W, (j + 1) W), W = VB / J, of the
// batch (populations <= W).  Four in ten non-empty voxels of cfg-2 (six in ten on cfg-3's 0.2 m grid) hold at most four points:
// a batch per voxel would run the loads, the index exchange and the emit for one to four rows.  perm[lo .. hi) in descending
// population; item i of the wave = voxels lo + (i groups + g) J + j.
template <bool FB, bool OB, bool ACC, bool FUSED, int VB, int J>
__device__ __forceinline__ void vp_vox_small(int lo, int hi, int gw, int nwaves, int cl, bool ingroup, int g, int gs, int groups, int lpr,
                                             __amdgpu_buffer_rsrc_t f_rsrc, __amdgpu_buffer_rsrc_t o_rsrc, unsigned row_in,
                                             unsigned lane_in, unsigned row_out, unsigned lane_out, unsigned pad_off, int kill,
                                             int *idw, float *prw, const VpVoxArgs &a) {
    constexpr int W = VB / J;
    const int n_items = (hi - lo + groups * J - 1) / (groups * J);
    const int myj = cl / W, myo = cl - myj * W;                   // (index lanes: cl < VB)
    for (int i = gw; i < n_items; i += nwaves) {                  // wave-uniform
        int vox[J], pb = 0, pe = 0, lenmax = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int vi = lo + (i * groups + g) * J + j;
            int4 rec = make_int4(-1, 0, 0, 0);
            if (ingroup && vi < hi) rec = a.perm[vi];
            vox[j] = rec.x;
            lenmax = max(lenmax, rec.z - rec.y);
            if (myj == j) { pb = rec.y; pe = rec.z; }
        }
        const int n = min(W, vp_group_max(lenmax, lpr, groups)); // the largest population of the item
        const int slot = pb + myo;
        int my_idx = -1;
        if (ingroup && cl < VB && slot < pe) my_idx = a.order[slot];
        float my_pr = 0.f;
        int my_row = my_idx;
        if constexpr (FUSED) {
            if (my_idx >= 0) {
                my_pr = a.prob[my_idx];
                const int b = vp_fast_div(my_idx, a.N, a.magN, a.shN);
                const int rem = my_idx - b * a.N;
                my_row = b * a.P + rem - vp_fast_div(rem, a.P, a.magP, a.shP) * a.P;
            }
        }
        if (ingroup && cl < VB) idw[g * VB + cl] = my_row;
        const vp_i32x4 *ip = reinterpret_cast<const vp_i32x4 *>(idw + gs * VB);
        int idx[VB];
#pragma unroll
        for (int q = 0; q < VB / 4; ++q) {
            const vp_i32x4 t = ip[q];
            idx[4 * q + 0] = t[0]; idx[4 * q + 1] = t[1]; idx[4 * q + 2] = t[2]; idx[4 * q + 3] = t[3];
        }
        float4 val[VB];
#pragma unroll
        for (int k = 0; k < VB; ++k) val[k] = vp_buf_load_row<FB>(f_rsrc, (a.dbg & 4) ? ~0u - 64u : (unsigned)idx[k] * row_in + lane_in);
        float pr[VB];
        if constexpr (FUSED) {
            if (ingroup && cl < VB) prw[g * VB + cl] = my_pr;
            const vp_f32x4 *pp = reinterpret_cast<const vp_f32x4 *>(prw + gs * VB);
#pragma unroll
            for (int q = 0; q < VB / 4; ++q) {
                const vp_f32x4 t = pp[q];
                pr[4 * q + 0] = t[0]; pr[4 * q + 1] = t[1]; pr[4 * q + 2] = t[2]; pr[4 * q + 3] = t[3];
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < W; ++o)
                if (o < n) vp_add_row<FUSED>(acc, val[j * W + o], FUSED ? pr[j * W + o] : 0.f);
            vp_buf_emit<OB, ACC>(o_rsrc, vox[j] | kill, row_out, lane_out, pad_off, acc);
        }
    }
}

template <bool FB, bool OB, bool ACC, bool FUSED, int VB>
__global__ __launch_bounds__(kBlock) void vp_gather_vox_kernel(const VpVoxArgs a) {
    if (a.gate != nullptr && *reinterpret_cast<const volatile int *>(a.gate) != 0) {
        if (a.zero_on_gate) vp_zero_output(a.out, a.out_bytes);
        return;
    }
    __shared__ __attribute__((aligned(16))) int idx_s[kBlock / 64][kMaxGroups * VB];
    __shared__ __attribute__((aligned(16))) float pr_s[kBlock / 64][FUSED ? kMaxGroups * VB : 4];
    __shared__ float4 red[kBlock / 64][64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int lpr = a.lpr, groups = a.groups, vb = a.vb;
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const bool ingroup = g < groups;
    // LDS block of the lane: its row group's; the lanes outside the groups (64 - groups * lpr of them) read group 0's block --
    // their rows are loaded and summed for nothing, never emitted
    const int gs = ingroup ? g : 0;
    const int src_cl = ingroup ? cl : 0;
    int *idw = idx_s[wid];
    float *prw = pr_s[wid];
    if (vb < VB) {                                               // (narrow rows only: entries k >= vb are never rewritten: dead)
        for (int i = lane; i < groups * VB; i += 64) {
            idw[i] = -1;
            if constexpr (FUSED) prw[i] = 0.f;
        }
    }
    const unsigned row_in = (unsigned)a.C * (FB ? 2u : 4u);
    const unsigned row_out = OB ? (unsigned)a.ldo * 2u : (unsigned)a.C * 4u;
    const unsigned lane_in = (unsigned)cl * (FB ? 8u : 16u);
    const unsigned lane_out = (unsigned)cl * (OB ? 8u : 16u);
    unsigned pad_off = ~0u;
    if constexpr (OB) {
        if (cl < ((a.ldo - a.C) >> 2)) pad_off = (unsigned)a.C * 2u + (unsigned)cl * 8u;
    }
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.feats), 0, (int)a.feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const int nwaves = (int)gridDim.x * (kBlock / 64);
    const int gw = (int)blockIdx.x * (kBlock / 64) + wid;
    const int nonempty = a.cls[0];
    const int n_mid_end = a.cls[32];                             // perm[0 .. n_mid_end): populations above 32 (up to 32: one row group)
    const int n_long = a.cls[31 + groups];                       // perm[0 .. n_long): above 32 * groups (classes step by 32 up to 320)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int kill = (a.dbg & 2) ? -1 : 0;                      // (probe: v | kill = -1 drops every row store)
    // populations 1 .. 4 and 5 .. 8 share a batch (four / two voxels per row group) when the batch has 16 slots
    const bool pack = VB == 16 && vb == 16 && !(a.dbg & 16);
    const int n8 = pack ? a.cls[8] : nonempty;                    // perm[n_mid_end .. n8): 9 .. 32 points (one voxel per row group)
    const int n4 = pack ? a.cls[4] : nonempty;                    // perm[n8 .. n4): 5 .. 8 points; perm[n4 .. nonempty): 1 .. 4
    // ---------------------------------------------------------------- rows of empty voxels (not in ACC mode)
    if (!ACC && !(a.dbg & 1)) {
        // perm's tail lists the empty voxels (in voxel order): eight records in flight per row group, then eight row stores.
        // (The alternative -- every row group walks a contiguous range of voxel ids and zeroes the empty ones it finds in
        // seg_start -- measured equal or slower everywhere but the fused form on cfg-3's grid, by 4 %.)
        constexpr int EU = 8;
        const int ngr = nwaves * groups;
        const int per = (a.V + ngr - 1) / ngr;
        const int v0 = (gw * groups + (ingroup ? g : 0)) * per;
        if (!(a.dbg & 8)) {                                       // the zero rows from perm's tail (probe bit 8: the walk below)
            const int n_rows = (a.V - nonempty + groups - 1) / groups;
            for (int i0 = gw * EU; i0 < n_rows; i0 += nwaves * EU) {
                int v[EU];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int vi = nonempty + (i0 + u) * groups + g;
                    v[u] = -1;
                    if (ingroup && i0 + u < n_rows && vi < a.V) v[u] = a.perm[vi].x;
                }
#pragma unroll
                for (int u = 0; u < EU; ++u) vp_buf_emit<OB, false>(o_rsrc, v[u], row_out, lane_out, pad_off, zero4);
            }
        } else
        for (int i0 = 0; i0 < per; i0 += EU) {                    // uniform trip count
            int s[EU + 1];
#pragma unroll
            for (int u = 0; u <= EU; ++u) {
                const int v = v0 + i0 + u;
                s[u] = v <= a.V ? a.seg_start[v] : 0;
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const int v = v0 + i0 + u;
                const bool emp = ingroup && i0 + u < per && v < a.V && s[u] == s[u + 1];
                vp_buf_emit<OB, false>(o_rsrc, emp ? v : -1, row_out, lane_out, pad_off, zero4);
            }
        }
    }

    // ---------------------------------------------------------------- the largest voxels: one workgroup each
    // Every phase deals its items with a static stride, and every phase continues where the previous one stopped (`woff`, in
    // waves): started at wave 0 each, the first few hundred workgroups got a large voxel, a middle one AND three small items in
    // a row while most of the grid only wrote zero rows.
    for (int i = (int)blockIdx.x; i < n_long; i += (int)gridDim.x) {      // block-uniform
        const int4 rec = a.perm[i];
        const int v = rec.x, b = rec.y, e = rec.z;
        const int np = (kBlock / 64) * groups;
        const int psz = (e - b + np - 1) / np;
        int pb = 0, pe = 0;
        if (ingroup) {
            pb = min(b + (wid * groups + g) * psz, e);
            pe = min(pb + psz, e);
        }
        float4 acc = zero4;
        vp_vox_piece<FB, FUSED, VB>(acc, pb, pe, psz, cl, ingroup, g, gs, vb, f_rsrc, row_in, lane_in, idw, prw, a);
        float4 wsum = zero4;
        for (int g2 = 0; g2 < groups; ++g2) vacc(wsum, vp_from_lane(acc, g2 * lpr + src_cl));
        if (g == 0) red[wid][cl] = wsum;
        __syncthreads();
        if (wid == 0) {
            float4 tot = red[0][src_cl];
#pragma unroll
            for (int w = 1; w < kBlock / 64; ++w) vacc(tot, red[w][src_cl]);
            vp_buf_emit<OB, ACC>(o_rsrc, (g == 0 ? v : -1) | kill, row_out, lane_out, pad_off, tot);
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- middle populations: one wave each, `groups` pieces
    int woff = (n_long % (int)gridDim.x) * (kBlock / 64);
    for (int i = n_long + (gw - woff + nwaves) % nwaves; i < n_mid_end; i += nwaves) {       // wave-uniform (empty when groups == 1)
        const int4 rec = a.perm[i];
        const int v = rec.x, b = rec.y, e = rec.z;
        const int psz = (e - b + groups - 1) / groups;
        int pb = 0, pe = 0;
        if (ingroup) {
            pb = min(b + g * psz, e);
            pe = min(pb + psz, e);
        }
        float4 acc = zero4;
        vp_vox_piece<FB, FUSED, VB>(acc, pb, pe, psz, cl, ingroup, g, gs, vb, f_rsrc, row_in, lane_in, idw, prw, a);
        float4 tot = zero4;
        for (int g2 = 0; g2 < groups; ++g2) vacc(tot, vp_from_lane(acc, g2 * lpr + src_cl));
        vp_buf_emit<OB, ACC>(o_rsrc, (g == 0 ? v : -1) | kill, row_out, lane_out, pad_off, tot);
    }
    // ---------------------------------------------------------------- small populations: one row group each
    const int n_small = (n8 - n_mid_end + groups - 1) / groups;
    woff = (woff + n_mid_end - n_long) % nwaves;
    for (int i = (gw - woff + nwaves) % nwaves; i < n_small; i += nwaves) {                  // wave-uniform
        const int vi = n_mid_end + i * groups + g;
        int v = -1, pb = 0, pe = 0;
        if (ingroup && vi < n8) {
            const int4 rec = a.perm[vi];
            v = rec.x; pb = rec.y; pe = rec.z;
        }
        const int maxlen = vp_group_max(pe - pb, lpr, groups);   // (items at a class border mix two populations)
        float4 acc = zero4;
        vp_vox_piece<FB, FUSED, VB>(acc, pb, pe, maxlen, cl, ingroup, g, gs, vb, f_rsrc, row_in, lane_in, idw, prw, a);
        vp_buf_emit<OB, ACC>(o_rsrc, v | kill, row_out, lane_out, pad_off, acc);
    }
    if constexpr (VB == 16) {
        if (pack) {
            woff = (woff + n_small) % nwaves;
            const int gw2 = (gw - woff + nwaves) % nwaves;
            woff = (woff + (n4 - n8 + groups * 2 - 1) / (groups * 2)) % nwaves;
            const int gw4 = (gw - woff + nwaves) % nwaves;
            vp_vox_small<FB, OB, ACC, FUSED, VB, 2>(n8, n4, gw2, nwaves, cl, ingroup, g, gs, groups, lpr, f_rsrc, o_rsrc, row_in, lane_in, row_out,
                                                    lane_out, pad_off, kill, idw, prw, a);
            vp_vox_small<FB, OB, ACC, FUSED, VB, 4>(n4, nonempty, gw4, nwaves, cl, ingroup, g, gs, groups, lpr, f_rsrc, o_rsrc, row_in, lane_in,
                                                    row_out, lane_out, pad_off, kill, idw, prw, a);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward gather  (voxel_pooling.py:58-69)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void vp_backward_kernel(long long total_elems, int C,
                                                             const int32_t *__restrict__ pos_memo,
                                                             const float *__restrict__ grad_out,
                                                             long long sb, long long sc, long long sy, long long sx,
                                                             float *__restrict__ grad_in) {
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= total_elems) return;
    const long long pt = idx / C;
    const int c = (int)(idx - pt * C);
    const int b = pos_memo[pt * 3 + 0];
    float g = 0.f;
    if (b != -1) {
        const int y = pos_memo[pt * 3 + 1], x = pos_memo[pt * 3 + 2];
        g = grad_out[b * sb + c * sc + y * sy + x * sx];
    }
    grad_in[idx] = g;
}

int check_common(int B, int N, int C, int X, int Y, int Z) {
    SGV3D_REQUIRE(B > 0 && N > 0 && C > 0 && X > 0 && Y > 0 && Z > 0,
                  "voxel_pooling: non-positive size (B=%d N=%d C=%d X=%d Y=%d Z=%d)", B, N, C, X, Y, Z);
    SGV3D_REQUIRE((long long)B * N < 0x7fffffffLL, "voxel_pooling: B*N=%lld does not fit int32 point ids",
                  (long long)B * N);
    SGV3D_REQUIRE((long long)B * Y * X < 0x7fffffffLL, "voxel_pooling: B*Y*X too large");
    return SGV3D_OK;
}

struct GatherGeom {
    bool v2;                // the slot-balanced gather covers this channel count
    int lpr, groups, ch;    // lanes per row, row groups per wave, slots per chunk
    int w_nom;              // slots between consecutive wave windows (window = groups * ch slots)
};

GatherGeom gather_geom(long long total_pts, int C) {
    GatherGeom G;
    G.v2 = false; G.lpr = G.groups = G.ch = G.w_nom = 0;
    if (C % 4 == 0 && C / 4 <= 64 && C / 4 >= 6) {
        G.v2 = true;
        G.lpr = C / 4;
        G.groups = 64 / G.lpr;
        G.ch = G.lpr - 2 < kChunkMax ? G.lpr - 2 : kChunkMax;
        static const int ch_env = [] { const char *e = getenv("SGV3D_VP_CH"); return e ? atoi(e) : 0; }();          // (probe knobs)
        static const int margin_env = [] { const char *e = getenv("SGV3D_VP_MARGIN"); return e ? atoi(e) : -1; }();
        if (ch_env > 0 && ch_env < G.ch) G.ch = ch_env;
        const int cap = G.groups * G.ch;
        // the margin lets the last owned run finish inside the window (same batch of loads) in most waves
        int margin = cap / 4 < 8 ? cap / 4 : 8;
        if (margin_env >= 0 && margin_env < cap) margin = margin_env;
        G.w_nom = cap - margin;
    }
    (void)total_pts;
    return G;
}

// workgroups appended to the gather grid for the plan's long runs (idle ones leave after one load)
constexpr int kLongBlocks = 1024;

// 0: by the rule in launch_gather, 1: slot-balanced kernel, 2: voxel-owner kernel (SGV3D_VP_KERNEL=slot | vox at load time,
// sgv3d_voxel_pooling_select_kernel at run time: tests and probes run both on the same data)
std::atomic<int> g_vp_kernel{[] { const char *e = getenv("SGV3D_VP_KERNEL"); return !e ? 0 : e[0] == 's' ? 1 : e[0] == 'v' ? 2 : 0; }()};

bool vp_use_vox(long long total_pts, long long V, bool fused) {
    // One kernel for both forms: the fused lift-splat launch is bitwise lift + operator only if they sum in the same order.
    // (On sparse grids -- cfg-3 batch 4: 2.8 points per voxel of the grid, nine voxels in ten empty -- the slot-balanced
    // kernel, which walks the voxels in spatial order, reads the operator's rows with more line reuse: 232 against 270 us.)
    (void)total_pts; (void)V; (void)fused;
    return g_vp_kernel.load(std::memory_order_relaxed) != 1;
}

template <bool FUSED, bool FB = false, bool OB = false, bool ACC = false>
int launch_gather(int B, int N, int C, int X, int Y, const void *plan, const float *feats,
                  const float *prob, const float *ctx, int P, float *out, void *workspace, size_t ws_bytes,
                  hipStream_t st, int ldo = 0, const int *gate = nullptr, bool zero_on_gate = false) {
    const PlanLayout L = plan_layout(B, N, X, Y);
    const char *base = static_cast<const char *>(plan);
    const int *seg = reinterpret_cast<const int *>(base + L.off_seg);
    const int *order = reinterpret_cast<const int *>(base + L.off_order);
    const int *slotvox = reinterpret_cast<const int *>(base + L.off_slotvox);
    const GatherGeom G = gather_geom(L.total, C);
    if (G.v2) {
        const long long waves = (L.total + G.w_nom - 1) / G.w_nom;
        const int nblk = cdiv(waves, kBlock / 64);
        const int nlong = L.long_cap < kLongBlocks ? L.long_cap : kLongBlocks;
        {
            // the instruction-lean kernel addresses both tensors with 32-bit byte offsets (dead slots point past the end)
            const unsigned long long fbytes = (FUSED ? (unsigned long long)B * P : (unsigned long long)L.total) * C * (FB ? 2 : 4);
            const unsigned long long obytes = (unsigned long long)L.V * (OB ? ldo * 2 : C * 4);
            static const bool generic_env = [] { const char *e = getenv("SGV3D_VP_GENERIC"); return e && e[0] == '1'; }();
            // The voxel-owner kernel (round 4) unless sgv3d_voxel_pooling_select_kernel / SGV3D_VP_KERNEL=slot asks for the
            // slot-balanced one of round 3: fused form 16 / 71 / 140 us against 21 / 138 / 166 us at cfg-2 / cfg-5 / cfg-3
            // batch 4, operator form 25.7 / 185 / 270 against 26.7 / 243 / 232 us.
            const bool use_vox = vp_use_vox(L.total, L.V, FUSED);
            if (fbytes < 0xfff00000ull && obytes < 0xfff00000ull && L.total < 0x7ff00000ll && !generic_env && use_vox) {
                VpVoxArgs a;
                a.order = order; a.seg_start = seg;
                a.perm = reinterpret_cast<const int4 *>(base + L.off_perm);
                a.cls = reinterpret_cast<const int *>(base + L.off_bins) + (size_t)kVoxClasses * L.ncw;
                a.feats = FUSED ? static_cast<const void *>(ctx) : static_cast<const void *>(feats); a.out = out; a.gate = gate;
                a.zero_on_gate = zero_on_gate ? 1 : 0;
                a.prob = prob; a.P = P;
                a.feat_bytes = (unsigned)fbytes; a.out_bytes = (unsigned)obytes;
                // rows in flight per lane: 16 (4 waves per SIMD) or, SGV3D_VP_VB=8, 8 (<= 76 VGPRs: 6-7 waves per SIMD);
                // SGV3D_VP_GRID: workgroups of the launch (probe knobs, tools/vp_probe3.py)
                static const int vb_env = [] { const char *e = getenv("SGV3D_VP_VB"); return e ? atoi(e) : 0; }();
                static const int grid_env = [] { const char *e = getenv("SGV3D_VP_GRID"); return e ? atoi(e) : 0; }();
                // SGV3D_VP_DEBUG: bits 8 / 16 pick other (equally correct) forms of the zero-row phase / small-voxel packing; bits
                // 1 / 2 / 4 drop the zero rows / row stores / row loads -- wrong results by design, for tools/vp_probe3.py only, and
                // honoured only together with SGV3D_VP_DEBUG_WRONG_RESULTS=1
                static const int dbg_env = [] {
                    const char *e = getenv("SGV3D_VP_DEBUG"), *w = getenv("SGV3D_VP_DEBUG_WRONG_RESULTS");
                    const int v = e ? atoi(e) : 0;
                    return (w && w[0] == '1') ? v : (v & ~7);
                }();
                a.dbg = dbg_env;
                const int VBsel = vb_env == 16 || vb_env == 8 ? vb_env : 16;
                // workgroups: 1 024 (4 per CU) for frames of up to a million points, 4 096 from two million (cfg-2: 15.3 us fused
                // with 1 024 against 16.7 / 18.5 with 2 048 / 4 096; cfg-3 batch 4: 238 us operator with 4 096 against 251 / 257;
                // cfg-5: 178 against 181 / 186)
                const int vgrid = grid_env > 0 ? grid_env : L.total >= 2000000 ? 4 * kVoxGrid : L.total >= 1000000 ? 2 * kVoxGrid : kVoxGrid;
                if (FUSED) {
                    const VpMagic mn = vp_magic(N), mp = vp_magic(P);
                    a.magN = mn.mag; a.shN = mn.sh; a.magP = mp.mag; a.shP = mp.sh;
                }
                a.V = (int)L.V; a.C = C; a.lpr = G.lpr; a.groups = G.groups; a.vb = G.lpr < VBsel ? G.lpr : VBsel;
                a.ldo = ldo; a.N = N;
                if (VBsel == 16)
                    hipLaunchKernelGGL((vp_gather_vox_kernel<FB, OB, ACC, FUSED, 16>), dim3(vgrid), dim3(kBlock), 0, st, a);
                else
                    hipLaunchKernelGGL((vp_gather_vox_kernel<FB, OB, ACC, FUSED, 8>), dim3(vgrid), dim3(kBlock), 0, st, a);
                return check_launch(FUSED ? "vp_gather_vox_kernel (lift-splat)" : "vp_gather_vox_kernel");
            }
            if (fbytes < 0xfff00000ull && obytes < 0xfff00000ull && L.total < 0x7ff00000ll && !generic_env) {
                VpFastArgs a;
                a.seg_start = seg; a.order = order; a.slot_voxel = slotvox;
                a.long_list = reinterpret_cast<const int *>(base + L.off_long);
                a.feats = FUSED ? static_cast<const void *>(ctx) : static_cast<const void *>(feats); a.out = out; a.gate = gate;
                a.zero_on_gate = zero_on_gate ? 1 : 0;
                a.prob = prob; a.P = P;
                a.feat_bytes = (unsigned)fbytes; a.out_bytes = (unsigned)obytes;
                a.V = (int)L.V; a.C = C; a.lpr = G.lpr; a.groups = G.groups; a.ch = G.ch; a.w_nom = G.w_nom; a.ldo = ldo;
                a.long_cap = L.long_cap; a.nblk_regular = nblk; a.N = N;
                hipLaunchKernelGGL((vp_gather_fast_kernel<FB, OB, ACC, FUSED>), dim3(nblk + nlong), dim3(kBlock), 0, st, a);
                return check_launch(FUSED ? "vp_gather_fast_kernel (lift-splat)" : "vp_gather_fast_kernel");
            }
        }
        hipLaunchKernelGGL((vp_gather3_kernel<FUSED, FB, OB, ACC>), dim3(nblk + nlong), dim3(kBlock), 0, st, L.V, C, G.lpr,
                           G.groups, G.ch, G.w_nom, seg, order, slotvox, feats, prob, ctx, N, P, out, ldo,
                           reinterpret_cast<const int *>(base + L.off_long), L.long_cap, nblk, gate, zero_on_gate ? 1 : 0);
        return check_launch(FUSED ? "vp_lift_splat(v3)" : "vp_gather3_kernel");
    }
    if constexpr (ACC) return fail(SGV3D_EINVAL, "voxel pooling: the accumulating gather needs 24 <= C <= 256, C %% 4 == 0 (got %d)", C);
    if (FB || OB) return fail(SGV3D_EINVAL, "voxel pooling: bf16 features / output need 24 <= C <= 256, C %% 4 == 0 (got %d)", C);
    const long long waves = (L.V + kVoxPerWave - 1) / kVoxPerWave;
    const int grid = cdiv(waves, kBlock / 64);
    if (C % 4 == 0) {
        const int ncols = C / 4;
        const int lpr = ncols < 64 ? ncols : 64;
        const int groups = 64 / lpr;
        hipLaunchKernelGGL((vp_gather_kernel<4, FUSED>), dim3(grid), dim3(kBlock), 0, st, L.V, C, lpr, groups,
                           seg, order, feats, prob, ctx, N, P, out);
    } else {
        const int lpr = C < 64 ? C : 64;
        const int groups = 64 / lpr;
        hipLaunchKernelGGL((vp_gather_kernel<1, FUSED>), dim3(grid), dim3(kBlock), 0, st, L.V, C, lpr, groups,
                           seg, order, feats, prob, ctx, N, P, out);
    }
    return check_launch(FUSED ? "vp_lift_splat_kernel" : "vp_gather_kernel");
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" size_t sgv3d_voxel_plan_bytes(int batch_size, int num_points, int num_voxel_x, int num_voxel_y) {
    if (batch_size <= 0 || num_points <= 0 || num_voxel_x <= 0 || num_voxel_y <= 0) return 0;
    return plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y).bytes;
}

namespace {

int plan_build_impl(int batch_size, int num_points, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                    const int32_t *geom_xyz, int32_t *pos_memo, void *plan, size_t plan_bytes, int sort_segments,
                    bool cached, hipStream_t st, const char *what, bool compare_done = false) {
    if (int rc = check_common(batch_size, num_points, 1, num_voxel_x, num_voxel_y, num_voxel_z)) return rc;
    SGV3D_REQUIRE(geom_xyz && plan, "%s: null pointer", what);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0, "%s: plan must be 16-B aligned", what);
    SGV3D_REQUIRE(!cached || compare_done || (reinterpret_cast<uintptr_t>(geom_xyz) & 15) == 0, "%s: geom_xyz must be 16-B aligned", what);
    const PlanLayout L = plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y);
    if (plan_bytes < L.bytes) return fail(SGV3D_ENOSPACE, "%s: plan has %zu bytes, needs %zu", what, plan_bytes, L.bytes);
    char *base = static_cast<char *>(plan);
    int *seg = reinterpret_cast<int *>(base + L.off_seg);
    int *cur = reinterpret_cast<int *>(base + L.off_cur);
    int *order = reinterpret_cast<int *>(base + L.off_order);
    int *blk = reinterpret_cast<int *>(base + L.off_blk);
    PlanHeader *hdr = reinterpret_cast<PlanHeader *>(base + L.off_hdr);
    int32_t *gcopy = reinterpret_cast<int32_t *>(base + L.off_geom);
    const int *dirty = cached ? &hdr->dirty : nullptr;
    const int p[7] = {kPlanMagic, batch_size, num_points, num_voxel_x, num_voxel_y, num_voxel_z, sort_segments ? 1 : 0};
    if (cached && !compare_done) {
        const long long n_ints = L.total * 3;
        // (<= 256 workgroups: every workgroup ends with a ticket atomic on one address; 1024 of them took 28 us for 11 MB)
        const int cgrid = (int)(cdiv(n_ints / 4 + 1, kBlock) < 256 ? cdiv(n_ints / 4 + 1, kBlock) : 256);
        hipLaunchKernelGGL(vp_geom_compare_kernel, dim3(cgrid), dim3(kBlock), 0, st, n_ints, geom_xyz, gcopy, hdr, p[0],
                           p[1], p[2], p[3], p[4], p[5], p[6]);
    }
    // (a plain kernel, not hipMemsetAsync: a memset node inside a captured hipGraph faulted on replay
    // -- "write access to a read-only page" -- once the host had made small allocations after the capture)
    hipLaunchKernelGGL(vp_zero_kernel, dim3(cdiv(L.V + 1, kBlock)), dim3(kBlock), 0, st, L.V + 1, cur, dirty);
    const int pgrid = cdiv(L.total, kBlock);
    hipLaunchKernelGGL(vp_count_kernel, dim3(pgrid), dim3(kBlock), 0, st, L.total, num_points, num_voxel_x,
                       num_voxel_y, num_voxel_z, geom_xyz, pos_memo, cur, dirty, cached ? gcopy : nullptr);
    hipLaunchKernelGGL(vp_scan_local_kernel, dim3(L.nblk), dim3(kBlock), 0, st, L.V, cur, seg, blk, dirty);
    int *long_list = reinterpret_cast<int *>(base + L.off_long);
    hipLaunchKernelGGL(vp_scan_top_kernel, dim3(1), dim3(kBlock), 0, st, L.nblk, blk, dirty, long_list);
    hipLaunchKernelGGL(vp_scan_add_kernel, dim3(cdiv(L.V + 1, kBlock)), dim3(kBlock), 0, st, L.V, L.nblk, blk,
                       seg, cur, dirty);
    hipLaunchKernelGGL(vp_long_list_kernel, dim3(cdiv(L.V, kBlock)), dim3(kBlock), 0, st, L.V, seg, long_list, L.long_cap, dirty);
    {   // voxels by population class, in voxel order inside a class (the voxel-owner gather's work list)
        int *tbl = reinterpret_cast<int *>(base + L.off_bins);
        int *cls = tbl + (size_t)kVoxClasses * L.ncw;
        int *perm = reinterpret_cast<int *>(base + L.off_perm);
        hipLaunchKernelGGL(vp_cls_count_kernel, dim3(L.ncw), dim3(kBlock), 0, st, L.V, seg, L.ncw, tbl, dirty);
        int *tot = cls + kVoxClasses;
        hipLaunchKernelGGL(vp_cls_scan_kernel, dim3(kVoxClasses), dim3(kBlock), 0, st, L.ncw, tbl, tot, dirty);
        hipLaunchKernelGGL(vp_cls_scatter_kernel, dim3(L.ncw), dim3(kBlock), 0, st, L.V, seg, L.ncw, tbl, tot, cls, perm, dirty);
    }
    hipLaunchKernelGGL(vp_fill_kernel, dim3(pgrid), dim3(kBlock), 0, st, L.total, num_points, num_voxel_x,
                       num_voxel_y, num_voxel_z, geom_xyz, cur, order, reinterpret_cast<int *>(base + L.off_slotvox), dirty, seg);
    if (sort_segments) {
        const int g_wave = (int)(L.V / 4 < 4096 ? (L.V + 3) / 4 : 4096);
        hipLaunchKernelGGL((vp_sort_wave_kernel<1>), dim3(g_wave), dim3(kBlock), 0, st, L.V, seg, order, 1, dirty);
        hipLaunchKernelGGL((vp_sort_wave_kernel<4>), dim3(g_wave), dim3(kBlock), 0, st, L.V, seg, order, 64, dirty);
        hipLaunchKernelGGL((vp_sort_wave_kernel<16>), dim3(g_wave), dim3(kBlock), 0, st, L.V, seg, order, 256, dirty);
        hipLaunchKernelGGL((vp_sort_wave_kernel<32>), dim3(g_wave), dim3(kBlock), 0, st, L.V, seg, order, 1024, dirty);
        const int g_large = (int)((L.V + 255) / 256 < 1024 ? (L.V + 255) / 256 : 1024);
        hipLaunchKernelGGL((vp_sort_segments_kernel<256, 8192>), dim3(g_large), dim3(256), 0, st, L.V, seg, order,
                           2048, 0x7fffffff, dirty);
    }
    if (cached)
        hipLaunchKernelGGL(vp_plan_commit_kernel, dim3(1), dim3(64), 0, st, hdr, p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
    return check_launch(what);
}

}  // namespace

extern "C" int sgv3d_voxel_plan_build(int batch_size, int num_points, int num_voxel_x, int num_voxel_y,
                                      int num_voxel_z, const int32_t *geom_xyz, int32_t *pos_memo,
                                      void *plan, size_t plan_bytes, int sort_segments, void *stream) {
    return plan_build_impl(batch_size, num_points, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz, pos_memo, plan,
                           plan_bytes, sort_segments, false, as_stream(stream), "voxel_plan_build");
}

extern "C" int sgv3d_voxel_plan_init(int batch_size, int num_points, int num_voxel_x, int num_voxel_y, void *plan,
                                     size_t plan_bytes, void *stream) {
    SGV3D_REQUIRE(plan && batch_size > 0 && num_points > 0 && num_voxel_x > 0 && num_voxel_y > 0, "voxel_plan_init: bad argument");
    const PlanLayout L = plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y);
    if (plan_bytes < L.bytes) return fail(SGV3D_ENOSPACE, "voxel_plan_init: plan has %zu bytes, needs %zu", plan_bytes, L.bytes);
    hipLaunchKernelGGL(vp_plan_init_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<PlanHeader *>(static_cast<char *>(plan) + L.off_hdr));
    return check_launch("voxel_plan_init");
}

extern "C" int sgv3d_voxel_plan_build_cached(int batch_size, int num_points, int num_voxel_x, int num_voxel_y,
                                             int num_voxel_z, const int32_t *geom_xyz, void *plan, size_t plan_bytes,
                                             int sort_segments, void *stream) {
    return plan_build_impl(batch_size, num_points, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz, nullptr, plan,
                           plan_bytes, sort_segments, true, as_stream(stream), "voxel_plan_build_cached");
}

// ================================================================================================
// Level-1 drop-in: the symbol the reference's pybind wrapper reaches (INTEGRATION.md level 1)
// ================================================================================================
namespace {

// compare geom_xyz with the copy the cached plan holds and write the pos_memo rows of the kept points ((b, y, x),
// voxel_pooling_forward_cuda.cu:25-28) in the same pass over geom_xyz.  A difference RAISES PlanHeader::dirty (and leaves a
// note for the host); nothing lowers it here -- the rebuild's commit kernel does -- so no workgroup has to know that it is
// the last one: the round-3 form ended every workgroup with a ticket atomic on one address (~25 ns each, serialised) and
// walked the tensors 12 bytes per lane and iteration, 12 us for 11 MB read.  Here a thread owns FOUR consecutive points =
// three 16-byte loads per tensor, all six in flight at once, one pass, no loop; rows of four kept points leave as three
// 16-byte stores.  VEC = false: tensors that are not 16-byte aligned (one point per thread).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void vp_level1_prologue_kernel(long long total_pts, int N, int X, int Y, int Z,
                                                                    const int32_t *__restrict__ geom,
                                                                    const int32_t *__restrict__ copy,
                                                                    int32_t *__restrict__ pos_memo, PlanHeader *__restrict__ hdr,
                                                                    int *__restrict__ host_flag) {
    bool diff = false;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if constexpr (VEC) {
        const long long pt0 = t * 4;
        if (pt0 + 3 < total_pts) {
            const int4 *g4 = reinterpret_cast<const int4 *>(geom) + t * 3;
            const int4 *c4 = reinterpret_cast<const int4 *>(copy) + t * 3;
            const int4 g0 = g4[0], g1 = g4[1], g2 = g4[2];
            const int4 c0 = c4[0], c1 = c4[1], c2 = c4[2];
            diff = (g0.x != c0.x) | (g0.y != c0.y) | (g0.z != c0.z) | (g0.w != c0.w) | (g1.x != c1.x) | (g1.y != c1.y) |
                   (g1.z != c1.z) | (g1.w != c1.w) | (g2.x != c2.x) | (g2.y != c2.y) | (g2.z != c2.z) | (g2.w != c2.w);
            if (pos_memo != nullptr) {
                const int px[4] = {g0.x, g0.w, g1.z, g2.y}, py[4] = {g0.y, g1.x, g1.w, g2.z}, pz[4] = {g0.z, g1.y, g2.x, g2.w};
                bool keep[4];
                int bb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    keep[i] = px[i] >= 0 && px[i] < X && py[i] >= 0 && py[i] < Y && pz[i] >= 0 && pz[i] < Z;
                    bb[i] = (int)((pt0 + i) / N);
                }
                if (keep[0] && keep[1] && keep[2] && keep[3]) {
                    int4 *m4 = reinterpret_cast<int4 *>(pos_memo) + t * 3;
                    m4[0] = make_int4(bb[0], py[0], px[0], bb[1]);
                    m4[1] = make_int4(py[1], px[1], bb[2], py[2]);
                    m4[2] = make_int4(px[2], bb[3], py[3], px[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (keep[i]) {
                            pos_memo[(pt0 + i) * 3 + 0] = bb[i];
                            pos_memo[(pt0 + i) * 3 + 1] = py[i];
                            pos_memo[(pt0 + i) * 3 + 2] = px[i];
                        }
                }
            }
        } else {
            for (long long pt = pt0; pt < total_pts; ++pt) {             // the last, partial quad
                const int x = geom[pt * 3 + 0], y = geom[pt * 3 + 1], z = geom[pt * 3 + 2];
                diff |= (x != copy[pt * 3 + 0]) | (y != copy[pt * 3 + 1]) | (z != copy[pt * 3 + 2]);
                if (pos_memo != nullptr && x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z) {
                    pos_memo[pt * 3 + 0] = (int)(pt / N);
                    pos_memo[pt * 3 + 1] = y;
                    pos_memo[pt * 3 + 2] = x;
                }
            }
        }
    } else {
        if (t < total_pts) {
            const int x = geom[t * 3 + 0], y = geom[t * 3 + 1], z = geom[t * 3 + 2];
            diff = (x != copy[t * 3 + 0]) | (y != copy[t * 3 + 1]) | (z != copy[t * 3 + 2]);
            if (pos_memo != nullptr && x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z) {
                pos_memo[t * 3 + 0] = (int)(t / N);
                pos_memo[t * 3 + 1] = y;
                pos_memo[t * 3 + 2] = x;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // re-arm the one-launch rebuild's barrier (it runs behind this kernel)
        hdr->bar = 0;
        hdr->err = 0;
    }
    if (__ballot(diff) != 0ull && (threadIdx.x & 63) == 0) {             // (rare: one store per wave that saw a difference)
        __hip_atomic_store(&hdr->dirty, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // sticky note to the host: a later call enqueues the (device-gated) rebuild when it sees it
        if (host_flag != nullptr) __hip_atomic_store(host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the atomic scatter as the fallback of a cached-plan call: runs only while *gate != 0 (geom_xyz is not the plan's)
__global__ __launch_bounds__(kBlock) void vp_atomic_gated_kernel(
    long long total_pts, int N, int C, int X, int Y, int Z, const int32_t *__restrict__ geom,
    const float *__restrict__ feats, float *__restrict__ out, const int *__restrict__ gate) {
    if (*reinterpret_cast<const volatile int *>(gate) == 0) return;
    // (a bounded grid walking the point blocks: in the common case -- gate clear -- the launch is a few hundred workgroups
    // that leave after one load, not one per 64 points)
    __shared__ int vid_s[kAtomicPts];
    const int tid = threadIdx.x;
    const long long nblocks = (total_pts + kAtomicPts - 1) / kAtomicPts;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {       // block-uniform trip count
        const long long p0 = blk * kAtomicPts;
        if (tid < kAtomicPts) {
            const long long pt = p0 + tid;
            int v = -1;
            if (pt < total_pts) {
                int x, y;
                v = voxel_of_point(geom, pt, (int)(pt / N), X, Y, Z, x, y);
            }
            vid_s[tid] = v;
        }
        __syncthreads();
        const long long left = (total_pts - p0) * C;
        const int nelem = (long long)kAtomicPts * C < left ? kAtomicPts * C : (int)left;
        const float *f = feats + (size_t)p0 * C;
        for (int e = tid; e < nelem; e += kBlock) {
            const int pl = e / C;
            const int v = vid_s[pl];
            if (v >= 0) __hip_atomic_fetch_add(out + (size_t)v * C + (e - pl * C), f[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

// Plans of the drop-in entry, one per (device, stream, sizes).  The C ABI otherwise owns no memory; this entry has to (the
// reference's signature has no room for a plan), so it is bounded (kLevel1Max entries, least recently used evicted),
// can be emptied (sgv3d_voxel_pooling_cache_clear) and switched off (SGV3D_VP_LEVEL1_CACHE=0: always the atomic scatter).
struct Level1Entry {
    int dev; hipStream_t st; int B, N, X, Y, Z;
    void *plan; size_t plan_bytes;
    int *host_flag;          // pinned, device-visible: set by the prologue kernel when geom_xyz is not the plan's
    int *flag_dev;           // the same word as the device addresses it
    bool need_build;         // host-side: enqueue the (device-gated) build with the next call
    bool pinned;             // recorded into a stream capture: a replayed graph reads and writes this plan, so it is never
                             // evicted and survives sgv3d_voxel_pooling_cache_clear (the graph may outlive any call we see)
    unsigned long long last_use;
};
constexpr int kLevel1Max = 8;
std::mutex g_l1_mutex;
std::vector<Level1Entry> g_l1;
unsigned long long g_l1_clock = 0;
unsigned long long g_l1_stats[4] = {0, 0, 0, 0};     // calls, planned launches, atomic-only calls, builds enqueued

bool level1_enabled() {
    static const bool v = [] { const char *e = getenv("SGV3D_VP_LEVEL1_CACHE"); return !(e && e[0] == '0'); }();
    return v;
}

void level1_free(Level1Entry &e) {
    // (best effort on a teardown path: the return codes have nobody to go to)
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(e.dev);
    if (e.plan) (void)hipFree(e.plan);
    if (e.host_flag) (void)hipHostFree(e.host_flag);
    (void)hipSetDevice(cur);
}

int launch_atomic(int B, int N, int C, int X, int Y, int Z, const int32_t *geom, const float *feats, float *out,
                  int32_t *pos_memo, hipStream_t st) {
    const long long total = (long long)B * N;
    hipLaunchKernelGGL(vp_atomic_kernel, dim3(cdiv(total, kAtomicPts)), dim3(kBlock), 0, st, total, N, C, X, Y, Z, geom, feats,
                       out, pos_memo);
    return check_launch("vp_atomic_kernel");
}

}  // namespace

extern "C" int sgv3d_voxel_pooling_forward_atomic(int batch_size, int num_points, int num_channels,
                                                  int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                                  const int32_t *geom_xyz, const float *input_features,
                                                  float *output_features, int32_t *pos_memo, void *stream) {
    if (int rc = check_common(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z)) return rc;
    SGV3D_REQUIRE(geom_xyz && input_features && output_features, "voxel_pooling_forward: null pointer");
    return launch_atomic(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz,
                         input_features, output_features, pos_memo, as_stream(stream));
}

namespace {

// the atomic scatter ADDS: a FRESH call (output_features holds garbage) zeroes the map first; ``gate`` != nullptr: only while
// *gate != 0 (the fallback of a cached-plan call, whose gather has written every row otherwise)
void zero_output(float *out, long long n, const int *gate, hipStream_t st) {
    hipLaunchKernelGGL(vp_zero_kernel, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, st, n, reinterpret_cast<int *>(out), gate);
}

// FRESH = false: the reference's contract (output_features pre-zeroed by the caller, sums are added to it,
// voxel_pooling.py:37-38 / voxel_pooling_forward_cuda.cu:31-34); FRESH = true: output_features is written, whatever it held
template <bool FRESH>
int level1_forward(int batch_size, int num_points, int num_channels, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                   const int32_t *geom_xyz, const float *input_features, float *output_features, int32_t *pos_memo,
                   void *stream) {
    const int B = batch_size, N = num_points, C = num_channels, X = num_voxel_x, Y = num_voxel_y, Z = num_voxel_z;
    if (int rc = check_common(B, N, C, X, Y, Z)) return rc;
    SGV3D_REQUIRE(geom_xyz && input_features && output_features, "voxel_pooling_forward: null pointer");
    hipStream_t st = as_stream(stream);
    const GatherGeom G = gather_geom((long long)B * N, C);
    const bool aligned = ((reinterpret_cast<uintptr_t>(input_features) | reinterpret_cast<uintptr_t>(output_features)) & 15) == 0;
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    g_l1_stats[0]++;
    const long long out_elems = (long long)B * Y * X * C;
    if (!level1_enabled() || !G.v2 || !aligned) {
        g_l1_stats[2]++;
        if (FRESH) zero_output(output_features, out_elems, nullptr, st);
        return launch_atomic(B, N, C, X, Y, Z, geom_xyz, input_features, output_features, pos_memo, st);
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Level1Entry *e = nullptr;
    for (auto &c : g_l1)
        if (c.dev == dev && c.st == st && c.B == B && c.N == N && c.X == X && c.Y == Y && c.Z == Z) e = &c;
    const PlanLayout L = plan_layout(B, N, X, Y);
    if (e == nullptr) {
        if (capturing) {            // no allocation inside a stream capture: the plain scatter is always right
            g_l1_stats[2]++;
            if (FRESH) zero_output(output_features, out_elems, nullptr, st);
            return launch_atomic(B, N, C, X, Y, Z, geom_xyz, input_features, output_features, pos_memo, st);
        }
        if ((int)g_l1.size() >= kLevel1Max) {
            size_t old = g_l1.size();
            for (size_t i = 0; i < g_l1.size(); ++i)
                if (!g_l1[i].pinned && (old == g_l1.size() || g_l1[i].last_use < g_l1[old].last_use)) old = i;
            if (old == g_l1.size()) {      // every plan is baked into some captured graph: serve this call without one
                g_l1_stats[2]++;
                if (FRESH) zero_output(output_features, out_elems, nullptr, st);
                return launch_atomic(B, N, C, X, Y, Z, geom_xyz, input_features, output_features, pos_memo, st);
            }
            level1_free(g_l1[old]);
            g_l1.erase(g_l1.begin() + old);
        }
        Level1Entry n{};
        n.dev = dev; n.st = st; n.B = B; n.N = N; n.X = X; n.Y = Y; n.Z = Z;
        n.plan_bytes = L.bytes;
        if (hipMalloc(&n.plan, L.bytes) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&n.host_flag), sizeof(int), hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            level1_free(n);
            g_l1_stats[2]++;
            if (FRESH) zero_output(output_features, out_elems, nullptr, st);
            return launch_atomic(B, N, C, X, Y, Z, geom_xyz, input_features, output_features, pos_memo, st);
        }
        *n.host_flag = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&n.flag_dev), n.host_flag, 0) != hipSuccess) n.flag_dev = nullptr;
        n.need_build = true;
        hipLaunchKernelGGL(vp_plan_init_kernel, dim3(1), dim3(kBlock), 0, st,
                           reinterpret_cast<PlanHeader *>(static_cast<char *>(n.plan) + L.off_hdr));
        g_l1.push_back(n);
        e = &g_l1.back();
    }
    e->last_use = ++g_l1_clock;
    if (capturing) e->pinned = true;
    char *base = static_cast<char *>(e->plan);
    PlanHeader *hdr = reinterpret_cast<PlanHeader *>(base + L.off_hdr);
    const int32_t *gcopy = reinterpret_cast<const int32_t *>(base + L.off_geom);
    // the host learns about a changed geom_xyz one or more calls late (it never waits for the device); until then such
    // calls take the gated scatter below, which is correct for any geom_xyz
    if (__atomic_load_n(e->host_flag, __ATOMIC_RELAXED) != 0) e->need_build = true;
    // (a call that enqueues the rebuild anyway needs no note: its prologue finds the stale plan different by construction)
    int *flag_dev = (e->need_build && !capturing) ? nullptr : e->flag_dev;
    const long long total = (long long)B * N;
    // (the plan's own copy of geom_xyz sits at a 256-byte aligned offset of a hipMalloc'ed block)
    const bool vec = ((reinterpret_cast<uintptr_t>(geom_xyz) | reinterpret_cast<uintptr_t>(pos_memo)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(vp_level1_prologue_kernel<true>, dim3(cdiv(cdiv(total, 4), kBlock)), dim3(kBlock), 0, st, total, N, X, Y,
                           Z, geom_xyz, gcopy, pos_memo, hdr, flag_dev);
    else
        hipLaunchKernelGGL(vp_level1_prologue_kernel<false>, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, st, total, N, X, Y, Z,
                           geom_xyz, gcopy, pos_memo, hdr, flag_dev);
    if (e->need_build && !capturing) {
        // from a stream: the multi-launch build (the whole chip per phase: 114 us at cfg-2), gated on the device -- its kernels
        // leave at once if the prologue found the plan current after all
        if (int rc = plan_build_impl(B, N, X, Y, Z, geom_xyz, nullptr, e->plan, e->plan_bytes, 1, true, st,
                                     "voxel_pooling_forward(level-1 plan)", /*compare_done=*/true)) return rc;
        e->need_build = false;
        __atomic_store_n(e->host_flag, 0, __ATOMIC_RELAXED);
        g_l1_stats[3]++;
    } else if (capturing) {
        // What a stream capture records: the rebuild as ONE gated launch (vp_plan_build_one_kernel: leaves at once while
        // geom_xyz is the plan's, lowers the dirty flag when it has rebuilt), then the gather and the scatter, both gated on
        // that flag: right for whatever geom_xyz a replay sees.  (Rounds 3-4 recorded the ~16 gated launches of the build
        // above: 27 us of empty launches per replay; a forked graph branch for them was slower still, 75 against 59.8 us.  The
        // price: a replay whose geom_xyz did change rebuilds on a fraction of the chip, ~0.33 ms instead of 0.11.)
        PlanOneArgs o;
        o.total = total; o.V = L.V; o.N = N; o.X = X; o.Y = Y; o.Z = Z; o.nblk = L.nblk; o.ncw = L.ncw; o.long_cap = L.long_cap;
        o.geom = geom_xyz;
        o.gcopy = reinterpret_cast<int32_t *>(base + L.off_geom);
        o.seg = reinterpret_cast<int *>(base + L.off_seg);
        o.cur = reinterpret_cast<int *>(base + L.off_cur);
        o.order = reinterpret_cast<int *>(base + L.off_order);
        o.slotvox = reinterpret_cast<int *>(base + L.off_slotvox);
        o.blk = reinterpret_cast<int *>(base + L.off_blk);
        o.long_list = reinterpret_cast<int *>(base + L.off_long);
        o.tbl = reinterpret_cast<int *>(base + L.off_bins);
        o.cls = o.tbl + (size_t)kVoxClasses * L.ncw;
        o.tot = o.cls + kVoxClasses;
        o.perm = reinterpret_cast<int *>(base + L.off_perm);
        o.hdr = hdr;
        const int pm[7] = {kPlanMagic, B, N, X, Y, Z, 1};
        for (int i = 0; i < 7; ++i) o.p[i] = pm[i];
        hipLaunchKernelGGL(vp_plan_build_one_kernel, dim3(vp_one_grid()), dim3(kBlock), 0, st, o);
        g_l1_stats[3]++;
    }
    g_l1_stats[1]++;
    // (FRESH: a gather that finds the plan stale zeroes the map on its way out -- the gated scatter below adds into zeros)
    if (int rc = launch_gather<false, false, false, !FRESH>(B, N, C, X, Y, e->plan, input_features, nullptr, nullptr, 1,
                                                            output_features, nullptr, 0, st, 0, &hdr->dirty, FRESH)) return rc;
    const long long ablocks = cdiv(total, kAtomicPts);
    hipLaunchKernelGGL(vp_atomic_gated_kernel, dim3((unsigned)(ablocks < 2048 ? ablocks : 2048)), dim3(kBlock), 0, st, total, N, C, X,
                       Y, Z, geom_xyz, input_features, output_features, &hdr->dirty);
    return check_launch("voxel_pooling_forward(level-1)");
}

}  // namespace

// The reference wrapper's call (src/voxel_pooling_forward.cpp:26-39): output_features pre-zeroed by the caller, added to.
extern "C" int sgv3d_voxel_pooling_forward(int batch_size, int num_points, int num_channels,
                                           int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                           const int32_t *geom_xyz, const float *input_features,
                                           float *output_features, int32_t *pos_memo, void *stream) {
    return level1_forward<false>(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz,
                                 input_features, output_features, pos_memo, stream);
}

// The same call for a caller that OWNS the output allocation (this build's Python operator): output_features is written --
// every row, empty voxels as zeros -- so the 4 Y X C byte zero fill the reference does first (voxel_pooling.py:37-38) is
// not needed.  Same plans, same sums, same pos_memo.
extern "C" int sgv3d_voxel_pooling_forward_fresh(int batch_size, int num_points, int num_channels,
                                                 int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                                 const int32_t *geom_xyz, const float *input_features,
                                                 float *output_features, int32_t *pos_memo, void *stream) {
    return level1_forward<true>(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz,
                                input_features, output_features, pos_memo, stream);
}

extern "C" int sgv3d_voxel_pooling_kernel_for(int batch_size, int num_points, int num_channels, int num_voxel_x, int num_voxel_y,
                                              int fused) {
    if (batch_size <= 0 || num_points <= 0 || num_voxel_x <= 0 || num_voxel_y <= 0) return 0;
    const GatherGeom G = gather_geom((long long)batch_size * num_points, num_channels);
    if (!G.v2) return 0;
    return vp_use_vox((long long)batch_size * num_points, (long long)batch_size * num_voxel_x * num_voxel_y, fused != 0) ? 2 : 1;
}

extern "C" int sgv3d_voxel_pooling_select_kernel(int which) {
    SGV3D_REQUIRE(which >= 0 && which <= 2, "voxel_pooling_select_kernel: 0 (rule), 1 (slot-balanced) or 2 (voxel-owner)");
    g_vp_kernel.store(which, std::memory_order_relaxed);
    return SGV3D_OK;
}

extern "C" int sgv3d_voxel_pooling_cache_clear(void) {
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    (void)hipDeviceSynchronize();
    std::vector<Level1Entry> kept;
    for (auto &e : g_l1) {
        if (e.pinned) kept.push_back(e);        // a captured graph holds its pointers (see Level1Entry::pinned)
        else level1_free(e);
    }
    g_l1.swap(kept);
    return SGV3D_OK;
}

extern "C" int sgv3d_voxel_pooling_cache_stats(unsigned long long *out4) {
    SGV3D_REQUIRE(out4 != nullptr, "voxel_pooling_cache_stats: null pointer");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    for (int i = 0; i < 4; ++i) out4[i] = g_l1_stats[i];
    return SGV3D_OK;
}

extern "C" size_t sgv3d_voxel_plan_stats_offset(int batch_size, int num_points, int num_voxel_x, int num_voxel_y) {
    if (batch_size <= 0 || num_points <= 0 || num_voxel_x <= 0 || num_voxel_y <= 0) return 0;
    return plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y).off_hdr;
}

extern "C" size_t sgv3d_voxel_pooling_workspace_bytes(int batch_size, int num_points, int num_channels) {
    if (batch_size <= 0 || num_points <= 0 || num_channels <= 0) return 0;
    return 16;      // the owner-computes gather needs none (round 2 staged partial rows here); kept for ABI compatibility
}

extern "C" int sgv3d_voxel_pooling_forward_planned(int batch_size, int num_points, int num_channels,
                                                   int num_voxel_x, int num_voxel_y, const void *plan,
                                                   const float *input_features, float *output_features,
                                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_common(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, 1)) return rc;
    SGV3D_REQUIRE(plan && input_features && output_features, "voxel_pooling_forward_planned: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(input_features) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(output_features) & 15) == 0,
                  "voxel_pooling_forward_planned: feature buffers must be 16-B aligned");
    return launch_gather<false>(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, plan,
                                input_features, nullptr, nullptr, 1, output_features, workspace, workspace_bytes,
                                as_stream(stream));
}

extern "C" int sgv3d_voxel_pooling_forward_planned_bf16(int batch_size, int num_points, int num_channels, int num_voxel_x,
                                                        int num_voxel_y, const void *plan, const void *input_features_bf16,
                                                        void *output_features, int out_bf16_ld, void *workspace,
                                                        size_t workspace_bytes, void *stream) {
    if (int rc = check_common(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, 1)) return rc;
    SGV3D_REQUIRE(plan && input_features_bf16 && output_features, "voxel_pooling_forward_planned_bf16: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(input_features_bf16) & 7) == 0 && (reinterpret_cast<uintptr_t>(output_features) & 15) == 0,
                  "voxel_pooling_forward_planned_bf16: features must be 8-B, output 16-B aligned");
    SGV3D_REQUIRE(out_bf16_ld == 0 || (out_bf16_ld >= num_channels && out_bf16_ld % 4 == 0 && out_bf16_ld - num_channels <= 4 * (num_channels / 4)),
                  "voxel_pooling_forward_planned_bf16: out_bf16_ld must be 0 (f32 output) or a multiple of 4 in [C, 2C]");
    if (out_bf16_ld)
        return launch_gather<false, true, true>(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, plan,
                                                static_cast<const float *>(input_features_bf16), nullptr, nullptr, 1,
                                                static_cast<float *>(output_features), workspace, workspace_bytes, as_stream(stream),
                                                out_bf16_ld);
    return launch_gather<false, true>(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, plan,
                                      static_cast<const float *>(input_features_bf16), nullptr, nullptr, 1,
                                      static_cast<float *>(output_features), workspace, workspace_bytes, as_stream(stream));
}

extern "C" int sgv3d_lift_splat_planned(int batch_size, int num_depth, int num_pixels, int num_channels,
                                        int num_voxel_x, int num_voxel_y, const void *plan, const float *prob,
                                        const float *context, float *output_features, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(num_depth > 0 && num_pixels > 0, "lift_splat_planned: non-positive size");
    const long long N = (long long)num_depth * num_pixels;
    SGV3D_REQUIRE(N < 0x7fffffffLL, "lift_splat_planned: D*P too large");
    if (int rc = check_common(batch_size, (int)N, num_channels, num_voxel_x, num_voxel_y, 1)) return rc;
    SGV3D_REQUIRE(plan && prob && context && output_features, "lift_splat_planned: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(context) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(output_features) & 15) == 0,
                  "lift_splat_planned: context/output must be 16-B aligned");
    return launch_gather<true>(batch_size, (int)N, num_channels, num_voxel_x, num_voxel_y, plan, nullptr, prob,
                               context, num_pixels, output_features, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int sgv3d_lift_splat_planned_bf16out(int batch_size, int num_depth, int num_pixels, int num_channels, int num_voxel_x,
                                                int num_voxel_y, const void *plan, const float *prob, const void *context,
                                                int context_bf16, void *output_bf16, int out_bf16_ld, void *workspace,
                                                size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(num_depth > 0 && num_pixels > 0, "lift_splat_planned_bf16out: non-positive size");
    const long long N = (long long)num_depth * num_pixels;
    SGV3D_REQUIRE(N < 0x7fffffffLL, "lift_splat_planned_bf16out: D*P too large");
    if (int rc = check_common(batch_size, (int)N, num_channels, num_voxel_x, num_voxel_y, 1)) return rc;
    SGV3D_REQUIRE(plan && prob && context && output_bf16, "lift_splat_planned_bf16out: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(context) & 15) == 0 && (reinterpret_cast<uintptr_t>(output_bf16) & 15) == 0,
                  "lift_splat_planned_bf16out: context/output must be 16-B aligned");
    SGV3D_REQUIRE(out_bf16_ld >= num_channels && out_bf16_ld % 4 == 0 && out_bf16_ld - num_channels <= 4 * (num_channels / 4),
                  "lift_splat_planned_bf16out: out_bf16_ld must be a multiple of 4 in [C, 2C]");
    if (context_bf16)      // context rows as bf16 (half the L2 -> CU bytes of the launch): products and sums stay f32
        return launch_gather<true, true, true>(batch_size, (int)N, num_channels, num_voxel_x, num_voxel_y, plan, nullptr, prob,
                                               static_cast<const float *>(context), num_pixels, static_cast<float *>(output_bf16),
                                               workspace, workspace_bytes, as_stream(stream), out_bf16_ld);
    return launch_gather<true, false, true>(batch_size, (int)N, num_channels, num_voxel_x, num_voxel_y, plan, nullptr, prob,
                                            static_cast<const float *>(context), num_pixels, static_cast<float *>(output_bf16), workspace,
                                            workspace_bytes, as_stream(stream), out_bf16_ld);
}

extern "C" int sgv3d_voxel_pooling_backward(int batch_size, int num_points, int num_channels,
                                            const int32_t *pos_memo, const float *grad_output, long long sb,
                                            long long sc, long long sy, long long sx, float *grad_input,
                                            void *stream) {
    SGV3D_REQUIRE(batch_size > 0 && num_points > 0 && num_channels > 0, "voxel_pooling_backward: non-positive size");
    SGV3D_REQUIRE(pos_memo && grad_output && grad_input, "voxel_pooling_backward: null pointer");
    const long long total = (long long)batch_size * num_points * num_channels;
    hipLaunchKernelGGL(vp_backward_kernel, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, as_stream(stream), total,
                       num_channels, pos_memo, grad_output, sb, sc, sy, sx, grad_input);
    return check_launch("vp_backward_kernel");
}
