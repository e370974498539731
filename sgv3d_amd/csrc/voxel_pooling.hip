// Voxel pooling ("splat") for gfx950 — the operator of ops/voxel_pooling (reference:
// ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36, ops/voxel_pooling/voxel_pooling.py:10-69).
//
// Three formulations, all HBM-bound indexing work (no contraction => no MFMA):
//   1. vp_atomic_kernel      reference-faithful scatter with float atomics.  One lane per (point,
//                            channel) so every atomic wave-instruction covers contiguous 256-B row
//                            segments (the shape the memory-side atomic units run at full rate);
//                            rows of dropped points are never read.
//   2. plan build + vp_gather_kernel   deterministic CSR formulation: count -> scan -> fill ->
//                            per-segment sort, then every output row is gathered with 16-B loads,
//                            reduced in registers by 64/LPR row groups per wave and written once.
//   3. vp_lift_splat_kernel  the same gather with rows formed on the fly as prob * context
//                            (never materialises the [B,N,C] lifted tensor).
#include "common.hpp"

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
    size_t off_seg, off_cur, off_order, off_slotvox, off_blk, off_hdr, off_geom, bytes;
};

// Cache header of a plan (sgv3d_voxel_plan_build_cached): which geom_xyz / grid the plan was built for.
struct PlanHeader {
    int dirty;       // result of the last compare: 1 = the build kernels of this call run, 0 = they return at once
    int diff;        // accumulator of the compare workgroups
    int ticket;      // last-workgroup election of the compare kernel
    int params[7];   // magic, B, N, X, Y, Z, sort_segments of the plan held (all 0 = none)
    int builds;      // number of real builds so far (statistics / tests)
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
    L.bytes = al(L.off_geom + sizeof(int) * 3 * (size_t)L.total);
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
        hdr->dirty = 1; hdr->diff = 0; hdr->ticket = 0; hdr->builds = 0;
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
}

__global__ __launch_bounds__(kBlock) void vp_zero_kernel(long long n, int *__restrict__ p, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(kBlock) void vp_count_kernel(long long total_pts, int N, int X, int Y, int Z,
                                                          const int32_t *__restrict__ geom,
                                                          int32_t *__restrict__ pos_memo,
                                                          int *__restrict__ count, const int *__restrict__ dirty,
                                                          int32_t *__restrict__ geom_copy) {
    VP_SKIP_IF_CLEAN(dirty);
    const long long pt = (long long)blockIdx.x * kBlock + threadIdx.x;
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

__global__ __launch_bounds__(kBlock) void vp_scan_local_kernel(long long V, const int *__restrict__ count,
                                                               int *__restrict__ seg_start,
                                                               int *__restrict__ blk_sum, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    __shared__ int wave_tot[kBlock / 64];
    const long long base = (long long)blockIdx.x * kScanElems + (long long)threadIdx.x * kScanPerThread;
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
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of blk_sum[0..nblk) in place; blk_sum[nblk] = grand total
__global__ __launch_bounds__(kBlock) void vp_scan_top_kernel(int nblk, int *__restrict__ blk_sum,
                                                             const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    __shared__ int wave_tot[kBlock / 64];
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

__global__ __launch_bounds__(kBlock) void vp_scan_add_kernel(long long V, int nblk,
                                                             const int *__restrict__ blk_sum,
                                                             int *__restrict__ seg_start,
                                                             int *__restrict__ cursor, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < V) {
        const int s = seg_start[i] + blk_sum[i / kScanElems];
        seg_start[i] = s;
        cursor[i] = s;
    } else if (i == V) {
        seg_start[V] = blk_sum[nblk];
    }
}

__global__ __launch_bounds__(kBlock) void vp_fill_kernel(long long total_pts, int N, int X, int Y, int Z,
                                                         const int32_t *__restrict__ geom,
                                                         int *__restrict__ cursor,
                                                         int *__restrict__ order,
                                                         int *__restrict__ slot_voxel, const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    const long long pt = (long long)blockIdx.x * kBlock + threadIdx.x;
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
        slot_voxel[slot] = v;
    }
}

// Segments of lo+1 .. 64*R points: one wave per voxel, the list lives in R registers per lane
// (element e = r*64 + lane) and is sorted by a bitonic network whose steps run over the cross-lane
// network (distance < 64) or between registers of one lane (distance >= 64): no LDS, no barrier, and
// every long voxel gets its own wave, so the per-voxel multiplicity skew costs no serialisation.
template <int R>
__global__ __launch_bounds__(kBlock) void vp_sort_wave_kernel(long long V, const int *__restrict__ seg_start,
                                                              int *__restrict__ order, int lo,
                                                              const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (kBlock / 64);
    // voxels are dealt to waves with a per-round rotation (97 is coprime to any power-of-two wave
    // count): long lists cluster in a few BEV columns and a plain stride would pile them on few waves
    for (long long it = 0; it * nwaves < V; ++it) {
        const long long v = it * nwaves + (wave0 + 97 * it) % nwaves;
        if (v >= V) continue;
        const int s = seg_start[v];
        const int n = seg_start[v + 1] - s;
        if (n <= lo || n > 64 * R) continue;   // wave-uniform
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
}

// Ascending sort of every segment whose length is in (lo, hi]: normalised bitonic network (all
// comparators ascending, virtual +inf padding => works for any length), data staged in LDS when it
// fits, otherwise sorted in place in global memory by the one workgroup that owns the segment.
template <int T, int LDS_CAP>
__global__ __launch_bounds__(T) void vp_sort_segments_kernel(long long V, const int *__restrict__ seg_start,
                                                             int *__restrict__ order, int lo, int hi,
                                                             const int *__restrict__ dirty) {
    VP_SKIP_IF_CLEAN(dirty);
    __shared__ int buf[LDS_CAP];
    __shared__ int todo[T];
    __shared__ int ntodo;
    const int tid = threadIdx.x;
    // The workgroup inspects T voxels at a time and queues the ones it has to sort.  Voxel ids are
    // dealt round-robin over the workgroups (v = slot * gridDim + block): long segments cluster in
    // neighbouring voxels (near the camera), a contiguous assignment would serialise them on one CU.
    const long long G = gridDim.x;
    for (long long sb = 0; sb * G < V; sb += T) {
        if (tid == 0) ntodo = 0;
        __syncthreads();
        {
            // rotate the column by 37 per row: a plain v = slot*G + block would hand one BEV column
            // (all rows of one x when G == X) to one workgroup, and the hot voxels sit in few columns
            const long long v = (sb + tid) * G + (blockIdx.x + 37 * (sb + tid)) % G;
            const int n = v < V ? seg_start[v + 1] - seg_start[v] : 0;
            if (n > lo && n <= hi) todo[atomicAdd(&ntodo, 1)] = tid;   // queue order does not matter
        }
        __syncthreads();
        const int nq = ntodo;
        for (int qi = 0; qi < nq; ++qi) {
            const long long v = (sb + todo[qi]) * G + (blockIdx.x + 37 * (sb + todo[qi])) % G;
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
// 2c. balanced gather ("v2").  Work is cut over the SORTED SLOTS, not over voxels: each LPR-lane row
// group owns `ch` consecutive slots of the plan's order[] (ch = min(16, LPR-2)), fetches all of
// their rows at once (ch x 16 B in flight per lane), and runs a segmented sum over them.  A run that
// covers a whole voxel is stored directly; a run cut by a chunk border goes to partial[chunk][0]
// (it started in an earlier chunk) or partial[chunk][1] (it starts here and continues), and
// vp_fixup_kernel adds the pieces of each cut voxel in ascending chunk order.  Every group does the
// same amount of work whatever the per-voxel multiplicity (mean 16, max 246 at cfg-2; max 775 on the
// 128^2 grid), which the one-wave-per-voxel kernel above cannot offer.  Deterministic.
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

// FB: the feature rows are bf16 (bf16 compute mode: sgv3d_lift_bf16 wrote them); sums stay f32.  OB: see vp_store_row.
template <bool FUSED, bool FB = false, bool OB = false>
__global__ __launch_bounds__(kBlock) void vp_gather2_kernel(
    long long V, int C, int lpr, int groups, int ch, const int *__restrict__ seg_start,
    const int *__restrict__ order, const int *__restrict__ slot_voxel, const float *__restrict__ feats,
    const float *__restrict__ prob, const float *__restrict__ ctx, int N, int P, float *__restrict__ out,
    float *__restrict__ partial, int ldo) {
    const int lane = threadIdx.x & 63;
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const int glane0 = g * lpr;
    const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long long chunk = wave * groups + g;
    const int T = seg_start[V];
    const long long base = chunk * ch;
    const bool active = g < groups && base < T;
    const int ncols = C >> 2;
    // lanes 0..ch+1 of the group fetch slot base-1+cl: voxel id (all) and point id (the ch inner ones)
    int my_vox = -1, my_idx = -1;
    if (active && cl < ch + 2) {
        const long long sl = base - 1 + cl;
        if (sl >= 0 && sl < T) {
            my_vox = slot_voxel[sl];
            if (cl >= 1 && cl <= ch) my_idx = order[sl];
        }
    }
    const int prev_vox = __shfl(my_vox, glane0, 64);
    const int next_vox = __shfl(my_vox, glane0 + ch + 1, 64);
    const int cnt = active ? (int)min((long long)ch, (long long)T - base) : 0;
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
        if (k < cnt && cl < ncols) {
            if constexpr (FUSED) {
                const int b = idx / N;
                const int pix = (idx - b * N) % P;
                pr[k] = prob[idx];
                val[k] = *reinterpret_cast<const float4 *>(ctx + ((size_t)b * P + pix) * C + (size_t)cl * 4);
            } else if constexpr (FB) {
                const vp_bf16x4 q = *reinterpret_cast<const vp_bf16x4 *>(reinterpret_cast<const __bf16 *>(feats) + (size_t)idx * C + (size_t)cl * 4);
                val[k] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
            } else {
                val[k] = *reinterpret_cast<const float4 *>(feats + (size_t)idx * C + (size_t)cl * 4);
            }
        }
    }
    // rows of empty voxels: every group of the launch clears its share (disjoint from every row the
    // gather / fix-up write, so no ordering is needed and no memset pass either)
    if (g < groups && cl < ncols) {
        const long long ngroups = (long long)gridDim.x * (kBlock / 64) * groups;
        const long long per = (V + ngroups - 1) / ngroups;
        const long long v0 = chunk * per, v1 = min(V, v0 + per);
        for (long long v = v0; v < v1; ++v)
            if (seg_start[v] == seg_start[v + 1]) vp_store_row<OB>(out, v, C, ldo, cl, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (!active || cl >= ncols) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur = vox[0];
    bool before = prev_vox == cur;       // the first run started in an earlier chunk
#pragma unroll
    for (int k = 0; k < kChunkMax; ++k) {
        if (k < cnt) {
            if (vox[k] != cur) {             // run of `cur` ended inside the chunk
                if (before) *reinterpret_cast<float4 *>(partial + ((size_t)chunk * 2 + 0) * C + (size_t)cl * 4) = acc;
                else vp_store_row<OB>(out, cur, C, ldo, cl, acc);
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
                cur = vox[k];
                before = false;
            }
            if constexpr (FUSED) vfma(acc, pr[k], val[k]);
            else vacc(acc, val[k]);
        }
    }
    const bool after = cnt == ch && next_vox == cur;   // the last run continues in the next chunk
    if (before) *reinterpret_cast<float4 *>(partial + ((size_t)chunk * 2 + 0) * C + (size_t)cl * 4) = acc;
    else if (after) *reinterpret_cast<float4 *>(partial + ((size_t)chunk * 2 + 1) * C + (size_t)cl * 4) = acc;
    else vp_store_row<OB>(out, cur, C, ldo, cl, acc);
}

// One row group per chunk: if the chunk's last run starts here and continues, it leads that voxel:
// out[v] = partial[j][1] + partial[j+1][0] + ... in ascending chunk order.  The leader test reads its
// four slot ids up front (independent loads) and the piece count comes from the voxel's segment
// bounds, so the partial rows are fetched as independent loads, not as a dependent chain.
template <bool OB = false>
__global__ __launch_bounds__(kBlock) void vp_fixup_kernel(long long V, int C, int lpr, int groups, int ch,
                                                          const int *__restrict__ seg_start,
                                                          const int *__restrict__ slot_voxel,
                                                          const float *__restrict__ partial, float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int g = lane / lpr;
    const int cl = lane - g * lpr;
    const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const long long j = wave * groups + g;
    const int T = seg_start[V];
    const long long base = j * ch;
    if (g >= groups || base + ch >= T || cl >= (C >> 2)) return;   // the last chunk cannot continue
    const int v_last = slot_voxel[base + ch - 1];
    const int v_next = slot_voxel[base + ch];
    const int v_first = slot_voxel[base];
    const int v_prev = base > 0 ? slot_voxel[base - 1] : -1;
    if (v_next != v_last) return;                                   // last run ends with this chunk
    if (v_first == v_last && v_prev == v_last) return;              // middle piece, not the leader
    const int v = v_last;
    const long long jl = ((long long)seg_start[v + 1] - 1) / ch;    // chunk holding the voxel's last point
    float4 sum = *reinterpret_cast<const float4 *>(partial + ((size_t)j * 2 + 1) * C + (size_t)cl * 4);
    for (long long jj = j + 1; jj <= jl; jj += 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            p[u] = (jj + u <= jl) ? *reinterpret_cast<const float4 *>(partial + ((size_t)(jj + u) * 2 + 0) * C + (size_t)cl * 4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (jj + u <= jl) vacc(sum, p[u]);
    }
    vp_store_row<OB>(out, v, C, ldo, cl, sum);
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
    bool v2;
    int lpr, groups, ch;
    long long nchunks;
};

GatherGeom gather_geom(long long total_pts, int C) {
    GatherGeom G;
    G.v2 = false; G.lpr = G.groups = G.ch = 0; G.nchunks = 0;
    if (C % 4 == 0 && C / 4 <= 64 && C / 4 >= 6) {
        G.v2 = true;
        G.lpr = C / 4;
        G.groups = 64 / G.lpr;
        G.ch = G.lpr - 2 < kChunkMax ? G.lpr - 2 : kChunkMax;
        G.nchunks = (total_pts + G.ch - 1) / G.ch;
    }
    return G;
}

template <bool FUSED, bool FB = false, bool OB = false>
int launch_gather(int B, int N, int C, int X, int Y, const void *plan, const float *feats,
                  const float *prob, const float *ctx, int P, float *out, void *workspace, size_t ws_bytes,
                  hipStream_t st, int ldo = 0) {
    const PlanLayout L = plan_layout(B, N, X, Y);
    const char *base = static_cast<const char *>(plan);
    const int *seg = reinterpret_cast<const int *>(base + L.off_seg);
    const int *order = reinterpret_cast<const int *>(base + L.off_order);
    const int *slotvox = reinterpret_cast<const int *>(base + L.off_slotvox);
    const GatherGeom G = gather_geom(L.total, C);
    if (G.v2) {
        const size_t need = sizeof(float) * (size_t)G.nchunks * 2 * C;
        if (!workspace || ws_bytes < need)
            return fail(SGV3D_ENOSPACE, "voxel pooling: workspace has %zu bytes, needs %zu", ws_bytes, need);
        const long long waves = (G.nchunks + G.groups - 1) / G.groups;
        const int grid = cdiv(waves, kBlock / 64);
        float *partial = static_cast<float *>(workspace);
        hipLaunchKernelGGL((vp_gather2_kernel<FUSED, FB, OB>), dim3(grid), dim3(kBlock), 0, st, L.V, C, G.lpr, G.groups, G.ch,
                           seg, order, slotvox, feats, prob, ctx, N, P, out, partial, ldo);
        hipLaunchKernelGGL(vp_fixup_kernel<OB>, dim3(grid), dim3(kBlock), 0, st, L.V, C, G.lpr, G.groups, G.ch, seg, slotvox,
                           partial, out, ldo);
        return check_launch(FUSED ? "vp_lift_splat(v2)" : "vp_gather2_kernel");
    }
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
extern "C" int sgv3d_voxel_pooling_forward(int batch_size, int num_points, int num_channels,
                                           int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                           const int32_t *geom_xyz, const float *input_features,
                                           float *output_features, int32_t *pos_memo, void *stream) {
    if (int rc = check_common(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z)) return rc;
    SGV3D_REQUIRE(geom_xyz && input_features && output_features, "voxel_pooling_forward: null pointer");
    const long long total = (long long)batch_size * num_points;
    const int grid = cdiv(total, kAtomicPts);
    hipLaunchKernelGGL(vp_atomic_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), total, num_points,
                       num_channels, num_voxel_x, num_voxel_y, num_voxel_z, geom_xyz, input_features,
                       output_features, pos_memo);
    return check_launch("vp_atomic_kernel");
}

extern "C" size_t sgv3d_voxel_plan_bytes(int batch_size, int num_points, int num_voxel_x, int num_voxel_y) {
    if (batch_size <= 0 || num_points <= 0 || num_voxel_x <= 0 || num_voxel_y <= 0) return 0;
    return plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y).bytes;
}

namespace {

int plan_build_impl(int batch_size, int num_points, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                    const int32_t *geom_xyz, int32_t *pos_memo, void *plan, size_t plan_bytes, int sort_segments,
                    bool cached, hipStream_t st, const char *what) {
    if (int rc = check_common(batch_size, num_points, 1, num_voxel_x, num_voxel_y, num_voxel_z)) return rc;
    SGV3D_REQUIRE(geom_xyz && plan, "%s: null pointer", what);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(plan) & 15) == 0, "%s: plan must be 16-B aligned", what);
    SGV3D_REQUIRE(!cached || (reinterpret_cast<uintptr_t>(geom_xyz) & 15) == 0, "%s: geom_xyz must be 16-B aligned", what);
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
    if (cached) {
        const long long n_ints = L.total * 3;
        const int cgrid = (int)(cdiv(n_ints / 4 + 1, kBlock) < 1024 ? cdiv(n_ints / 4 + 1, kBlock) : 1024);
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
    hipLaunchKernelGGL(vp_scan_top_kernel, dim3(1), dim3(kBlock), 0, st, L.nblk, blk, dirty);
    hipLaunchKernelGGL(vp_scan_add_kernel, dim3(cdiv(L.V + 1, kBlock)), dim3(kBlock), 0, st, L.V, L.nblk, blk,
                       seg, cur, dirty);
    hipLaunchKernelGGL(vp_fill_kernel, dim3(pgrid), dim3(kBlock), 0, st, L.total, num_points, num_voxel_x,
                       num_voxel_y, num_voxel_z, geom_xyz, cur, order, reinterpret_cast<int *>(base + L.off_slotvox), dirty);
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

extern "C" size_t sgv3d_voxel_plan_stats_offset(int batch_size, int num_points, int num_voxel_x, int num_voxel_y) {
    if (batch_size <= 0 || num_points <= 0 || num_voxel_x <= 0 || num_voxel_y <= 0) return 0;
    return plan_layout(batch_size, num_points, num_voxel_x, num_voxel_y).off_hdr;
}

extern "C" size_t sgv3d_voxel_pooling_workspace_bytes(int batch_size, int num_points, int num_channels) {
    if (batch_size <= 0 || num_points <= 0 || num_channels <= 0) return 0;
    const GatherGeom G = gather_geom((long long)batch_size * num_points, num_channels);
    return G.v2 ? sizeof(float) * (size_t)G.nchunks * 2 * num_channels : 16;
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
