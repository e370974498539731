// CenterHead branches fused, F(4x4, 3x3) form: [3x3 64 -> 64 + BN + ReLU] and [3x3 64 -> c_k + bias] of all branches
// (mmdet3d SeparateHead, reached through layers/heads/bev_height_head.py:110) in one kernel -- the same contract as
// conv_wino_head_kernel (conv_wino.hip), with 36 / 16 / 9 = 1/4 of the direct form's multiplications instead of 1/2.25.
//
// Why this shape.  Every branch's first layer reads the SAME 64-channel shared map, so the transformed input
// V = B^T d B of a workgroup's 16x16-pixel block (16 tiles x 36 positions x 64 channels x 4 B = 144 KB) is computed ONCE,
// kept in LDS for the whole kernel and is the B operand of every MFMA of all 36 branches: no input transform per branch
// (the F(2x2) kernel recomputes it 36 times: 0.15 M vector cycles per SIMD and frame), no patch traffic in the main loop.
// The fp32 frame is issue-bound (MFMA cycles + 4 x vector instructions), so what counts is the number of either.
//
//   MFMA      v_mfma_f32_16x16x4_f32, M = 16 hidden channels (A = U = G g G^T, streamed from L2 straight into registers in
//             fragment order), N = the block's 16 tiles (B = V from LDS), K = 64 input channels.  A wave owns 16 hidden
//             channels of the branch and ALL 36 positions: 36 accumulators x 4 registers, so the output transform
//             A^T M A of a (tile, channel) happens inside one lane -- no exchange of accumulators.
//             lane (g = lane / 16, j = lane % 16): tile j = 4 tx + ty, hidden channels 16 wave + 4 g + (0..3).
//   epilogue  BN + ReLU in registers (the hidden map never exists in memory, not even LDS).  The final 3x3 convolution
//             runs in SCATTER form: a lane adds its 4 channels x 16 hidden pixels into the 6x6 output neighbourhood of its
//             tile (576 FMAs per output channel, the same count as the gather form), then
//               * overlap-add between the 16 tile-lanes of a row through DPP (separable: horizontal row_shr/shl:4, then
//                 vertical row_shr/shl:1), which also completes the partial sums owed to the one-pixel ring around the
//                 block (top / bottom / left / right, the layout head_ring_fixup_kernel of conv_wino.hip consumes),
//               * reduce-scatter over the wave's 4 channel groups with v_permlane32_swap / v_permlane16_swap
//                 (2 values per swap: 16 + 20 values -> 4 + 5 per lane in 27 swaps + 27 adds),
//               * the four waves' sums meet in 9 KB of LDS and are added in fixed order.  Deterministic, no atomics.
//   LDS       V 147 456 B (XOR-swizzled 16-byte slots: conflict-free for the transform's writes and the fragment reads)
//             + 9 216 B for the cross-wave sums = 156 672 B; one workgroup (4 waves) per CU.
//   L2        a workgroup streams the branch's 590 KB of transformed weights per branch (each wave its 16 channels'
//             147 KB, in the order it consumes them, 9 fragments = 1 150 MFMA cycles ahead); all workgroups walk the
//             branches in step, so the 21 MB stay L2 hits.
#include "conv_common.hpp"

using namespace sgv3d;

namespace sgv3d {
// conv_wino.hip: adds the ring partial sums of the neighbouring blocks to the border pixels, fixed order
int launch_head_ring_fixup(int batch, int h, int w, int total_out, const float *ring, float *out, hipStream_t st);
}  // namespace sgv3d

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H4_POS = 36;
constexpr int H4_V_SLOTS = H4_POS * 16 * 16;               // 16-byte slots: [pos][tile][channel quad ^ tile]
constexpr int H4_V_BYTES = H4_V_SLOTS * 16;                // 147456
constexpr int H4_RED = 9;                                  // floats per lane in the cross-wave area: 4 inner + 5 ring
constexpr int H4_RED_BYTES = 4 * 64 * H4_RED * 4;          // 9216
constexpr int H4_LDS = H4_V_BYTES + H4_RED_BYTES;
constexpr int H4_FRAG = 64 * 16;                           // bytes of one (k-quad, position) weight fragment of a wave
constexpr int H4_BRANCH = 4 * H4_POS * H4_FRAG;            // bytes of one (channel chunk, branch): 147456
constexpr int H4_AHEAD = 9;                                // weight fragments in flight per wave (divides 36)
constexpr int H4_RING = 68;                                // ring pixels around a 16x16 block (conv_wino.hip: HEAD_RING)

struct Head4Args {
    const float *x;               // NHWC shared map
    const float *u;               // [chunk 4][branch][k-quad 4][pos 36][lane 64][4]
    const float *scale1, *bias1;  // folded BN of the hidden layers [nb * 64] (null: 1 / 0)
    const float *w2, *bias2;      // [total_out][3][3][64], [total_out]
    const int *out_begin;         // [nb + 1]
    float *out;                   // NCHW [batch][total_out][h][w]
    float *ring;                  // [blocks][total_out][68]
    int x_ld, x_coff, h, w, nb, total_out, wb_y, wb_x;
    unsigned u_bytes;
};

// B^T (6 -> 6) and A^T (6 -> 4) of F(4x4, 3x3), interpolation points 0, +-1, +-2, inf (the constants of conv_wino4.hip)
template <typename T>
__device__ __forceinline__ void h4_bt(T &d0, T &d1, T &d2, T &d3, T &d4, T &d5) {
    const T a = d4 - 4.f * d2, b = d3 - 4.f * d1;
    const T c = d4 - d2, e = d3 - d1;
    const T t0 = 4.f * d0 + (d4 - 5.f * d2);
    const T t5 = 4.f * d1 + (d5 - 5.f * d3);
    d0 = t0;
    d1 = a + b;
    d2 = a - b;
    d3 = c + 2.f * e;
    d4 = c - 2.f * e;
    d5 = t5;
}
__device__ __forceinline__ void h4_at(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1,
                                      float &y2, float &y3) {
    const float p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4;
    y0 = (m0 + p) + r;
    y1 = fmaf(2.f, s, q);
    y2 = fmaf(4.f, r, p);
    y3 = fmaf(8.f, s, q) + m5;
}

__device__ __forceinline__ f32x4 h4_wload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// value of lane (l - N) / (l + N) of the 16-lane row, 0 where that lane is outside the row
template <int CTRL>
__device__ __forceinline__ float h4_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110;

#define H4_SB() __builtin_amdgcn_sched_barrier(0)
// every wave's LDS accesses have landed -> barrier; global loads (the weight fragments in flight) and stores stay in flight
#define H4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// a, b: two values per lane.  Returns in lanes [0, 32) a(l) + a(l + 32), in lanes [32, 64) b(l - 32) + b(l).
// Inline assembly: the second result of __builtin_amdgcn_permlane32_swap / 16_swap comes back as a copy of the first with this
// ROCm's hipcc (tools/ubench/lane_semantics.hip prints both); s_nop 1 is the wait the compiler itself puts in front of the swap
// after a vector write of its operands.
__device__ __forceinline__ float h4_fold32(float a, float b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}
// rows 0 / 2: a(row) + a(row + 1); rows 1 / 3: b(row - 1) + b(row)
__device__ __forceinline__ float h4_fold16(float a, float b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}

// One k-quad (16 input channels) of all 36 positions: 144 MFMAs.  vb: the lane's slot of position 0 for this k-quad.
template <bool FIRST>
__device__ __forceinline__ void h4_kquad(f32x4 (&acc)[H4_POS], f32x4 (&wf)[H4_AHEAD], const f32x4 *vb,
                                         __amdgpu_buffer_rsrc_t rsrc, unsigned w_lane, unsigned w_off) {
    // two positions per group, their MFMAs alternating: consecutive MFMAs never wait for each other's accumulator
    f32x4 bv0 = vb[0], bv1 = vb[256];
#pragma unroll
    for (int p = 0; p < H4_POS; p += 2) {
        // the scheduler would otherwise sink the weight loads to just in front of their use (two in flight instead of nine)
        H4_SB();
        const f32x4 bn0 = vb[(p + 2 < H4_POS ? p + 2 : p) * 256], bn1 = vb[(p + 3 < H4_POS ? p + 3 : p + 1) * 256];
        const f32x4 av0 = wf[p % H4_AHEAD], av1 = wf[(p + 1) % H4_AHEAD];
        wf[p % H4_AHEAD] = h4_wload(rsrc, w_lane, w_off + (unsigned)((p + H4_AHEAD) * H4_FRAG));
        wf[(p + 1) % H4_AHEAD] = h4_wload(rsrc, w_lane, w_off + (unsigned)((p + 1 + H4_AHEAD) * H4_FRAG));
        H4_SB();
        if constexpr (FIRST) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv0.x, z, 0, 0, 0);
            acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv1.x, z, 0, 0, 0);
        } else {
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv0.x, acc[p], 0, 0, 0);
            acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv1.x, acc[p + 1], 0, 0, 0);
        }
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv0.y, acc[p], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv1.y, acc[p + 1], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.z, bv0.z, acc[p], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.z, bv1.z, acc[p + 1], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.w, bv0.w, acc[p], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.w, bv1.w, acc[p + 1], 0, 0, 0);
        H4_SB();
        bv0 = bn0;
        bv1 = bn1;
    }
}

// V = B^T d B of the 16 tiles of the block at (oy0, ox0) for 64 input channels starting at xb -> LDS, in two halves so that a
// caller can put work between the loads and their use.  Thread = (tile, channel quad): consecutive threads read consecutive
// 16-byte pieces of a pixel's channel row.
__device__ __forceinline__ void h4_load_raw(f32x4 (&d)[6][6], const float *xb, int x_ld, int h, int w, int oy0, int ox0, int tid) {
    const int jt = tid >> 4, cq = tid & 15;
    const int iy0 = oy0 + 4 * (jt & 3) - 1, ix0 = ox0 + 4 * (jt >> 2) - 1;
    xb += cq * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int iy = iy0 + i;
        const bool rok = (unsigned)iy < (unsigned)h;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const int ix = ix0 + jj;
            const bool ok = rok & ((unsigned)ix < (unsigned)w);
            d[i][jj] = ok ? *reinterpret_cast<const f32x4 *>(xb + ((size_t)iy * w + ix) * x_ld) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}
__device__ __forceinline__ void h4_store_v(f32x4 *smem, f32x4 (&d)[6][6], int tid) {
    const int jt = tid >> 4, cq = tid & 15;
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) h4_bt(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj]);
    f32x4 *const vw = smem + jt * 16 + (cq ^ jt);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        h4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) vw[(i * 6 + jj) * 256] = d[i][jj];
    }
}
__device__ __forceinline__ void h4_build_v(f32x4 *smem, const float *xb, int x_ld, int h, int w, int oy0, int ox0, int tid) {
    f32x4 d[6][6];
    h4_load_raw(d, xb, x_ld, h, w, oy0, ox0, tid);
    h4_store_v(smem, d, tid);
}

__global__ __launch_bounds__(256, 1) void head_wino4_kernel(const Head4Args a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    float *const red = reinterpret_cast<float *>(smem + H4_V_SLOTS);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x;
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * 16, ox0 = bx * 16;

    // ---- V = B^T d B of the block's 16 tiles, once -------------------------------------------------------------------
    h4_build_v(smem, a.x + (size_t)img * a.h * a.w * a.x_ld + a.x_coff, a.x_ld, a.h, a.w, oy0, ox0, tid);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.u, 0, (int)a.u_bytes, 0x00020000);
    const unsigned w_lane = (unsigned)lane * 16u;
    unsigned w_off = (unsigned)wave * (unsigned)a.nb * (unsigned)H4_BRANCH;
    f32x4 wf[H4_AHEAD];
#pragma unroll
    for (int r = 0; r < H4_AHEAD; ++r) wf[r] = h4_wload(rsrc, w_lane, w_off + (unsigned)(r * H4_FRAG));
    const int g = lane >> 4, j = lane & 15;
    __syncthreads();

    const size_t plane = (size_t)a.h * a.w;
    for (int br = 0; br < a.nb; ++br) {
        f32x4 acc[H4_POS];
        h4_kquad<true>(acc, wf, smem + j * 16 + (g ^ j), rsrc, w_lane, w_off);
#pragma unroll 1
        for (int q = 1; q < 4; ++q)
            h4_kquad<false>(acc, wf, smem + j * 16 + ((4 * q + g) ^ j), rsrc, w_lane, w_off + (unsigned)(q * H4_POS * H4_FRAG));
        w_off += (unsigned)H4_BRANCH;

        // ---- epilogue.  Coordinates pass through an opaque copy so that nothing derived from them is hoisted in front
        // of the branch loop and carried (or spilled) across the main loop.
        int tid_ = tid, oy0_ = oy0, ox0_ = ox0;
        asm volatile("" : "+v"(tid_), "+s"(oy0_), "+s"(ox0_));
        const int lane_ = tid_ & 63, g_ = lane_ >> 4, j_ = lane_ & 15;
        const int tx = j_ >> 2, ty = j_ & 3;
        const int ch0 = br * 64 + wave * 16 + 4 * g_;
        const f32x4 sc = a.scale1 ? *reinterpret_cast<const f32x4 *>(a.scale1 + ch0) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = a.bias1 ? *reinterpret_cast<const f32x4 *>(a.bias1 + ch0) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int o0 = __builtin_amdgcn_readfirstlane(a.out_begin[br]);
        const int cb = __builtin_amdgcn_readfirstlane(a.out_begin[br + 1]) - o0;
        const bool full = (oy0_ + 16 <= a.h) & (ox0_ + 16 <= a.w);
        // hidden pixels of the lane's tile, 4 channels: A^T M A, BN, ReLU; zero outside the image
        float hd[4][16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t[4][6];
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
                h4_at(acc[jj][r], acc[6 + jj][r], acc[12 + jj][r], acc[18 + jj][r], acc[24 + jj][r], acc[30 + jj][r], t[0][jj],
                      t[1][jj], t[2][jj], t[3][jj]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float y[4];
                h4_at(t[u][0], t[u][1], t[u][2], t[u][3], t[u][4], t[u][5], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int v = 0; v < 4; ++v) hd[r][u * 4 + v] = fmaxf(fmaf(y[v], sc[r], sh[r]), 0.f);
            }
        }
        if (!full) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool in = (oy0_ + 4 * ty + u < a.h) & (ox0_ + 4 * tx + v < a.w);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hd[r][u * 4 + v] = in ? hd[r][u * 4 + v] : 0.f;
                }
        }
        const bool up_ok = ty > 0, dn_ok = ty < 3;
#pragma unroll 1
        for (int c2 = 0; c2 < cb; ++c2) {
            // final 3x3 convolution, scatter form: c[oy + 1][ox + 1], (oy, ox) in [-1, 4]^2 relative to the tile
            const float *wp = a.w2 + (size_t)(o0 + c2) * 576 + wave * 16 + 4 * g_;
            f32x4 wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4 *>(wp + t * 64);
            float c[6][6];
#pragma unroll
            for (int y = 0; y < 6; ++y)
#pragma unroll
                for (int x = 0; x < 6; ++x) c[y][x] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int py = 0; py < 4; ++py)
#pragma unroll
                    for (int px = 0; px < 4; ++px)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
                                c[py - ky + 2][px - kx + 2] = fmaf(wv[ky * 3 + kx][r], hd[r][py * 4 + px], c[py - ky + 2][px - kx + 2]);
            // overlap-add between the tiles of the block: horizontal (tile j -+ 4), then vertical (tile j -+ 1)
#pragma unroll
            for (int y = 0; y < 6; ++y) {
                const float fl = h4_dpp<DPP_ROW_SHR + 4>(c[y][5]), fr = h4_dpp<DPP_ROW_SHL + 4>(c[y][0]);
                c[y][1] += fl;
                c[y][4] += fr;
            }
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                const float fu = h4_dpp<DPP_ROW_SHR + 1>(c[5][x]), fd = h4_dpp<DPP_ROW_SHL + 1>(c[0][x]);
                c[1][x] += up_ok ? fu : 0.f;
                c[4][x] += dn_ok ? fd : 0.f;
            }
            // sum over the wave's 16 channels: lane (g', j) ends with inner row g' of its tile and ring values 5 g' .. + 4
            float s8[8], f4v[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) s8[i] = h4_fold32(c[1 + (i >> 2)][1 + (i & 3)], c[3 + (i >> 2)][1 + (i & 3)]);
#pragma unroll
            for (int i = 0; i < 4; ++i) f4v[i] = h4_fold16(s8[i], s8[4 + i]);
            // ring values of the tile, in the order top c[0][0..5], bottom c[5][0..5], left c[1..4][0], right c[1..4][5]
            float rv[20];
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                rv[x] = c[0][x];
                rv[6 + x] = c[5][x];
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                rv[12 + y] = c[1 + y][0];
                rv[16 + y] = c[1 + y][5];
            }
            float s10[10], f5[5];
#pragma unroll
            for (int i = 0; i < 10; ++i) s10[i] = h4_fold32(rv[i], rv[10 + i]);
#pragma unroll
            for (int i = 0; i < 5; ++i) f5[i] = h4_fold16(s10[i], s10[5 + i]);
            H4_BARRIER();                                      // the previous round's readers are done
            {
                float *rp = red + (wave * 64 + lane_) * H4_RED;
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[i] = f4v[i];
#pragma unroll
                for (int i = 0; i < 5; ++i) rp[4 + i] = f5[i];
            }
            H4_BARRIER();
            {   // own pixels: thread = (lane L of the reduce layout, x)
                const int L = tid_ >> 2, i = tid_ & 3;
                const int gq = L >> 4, jl = L & 15;
                const float *rp = red + L * H4_RED + i;
                const float s = ((rp[0] + rp[64 * H4_RED]) + rp[128 * H4_RED]) + rp[192 * H4_RED];
                const int Y = oy0_ + 4 * (jl & 3) + gq, X = ox0_ + 4 * (jl >> 2) + i;
                if (Y < a.h && X < a.w) a.out[((size_t)img * a.total_out + o0 + c2) * plane + (size_t)Y * a.w + X] = s + a.bias2[o0 + c2];
            }
            if (tid_ < H4_RING) {   // ring slot -> (tile, value) -> (lane, slot) of the reduce layout
                int jl, k;
                if (tid_ < 36) {
                    const int r = tid_ < 18 ? tid_ : tid_ - 18, tyr = tid_ < 18 ? 0 : 3;
                    const int txr = r == 0 ? 0 : r == 17 ? 3 : (r - 1) >> 2;
                    k = (r == 0 ? 0 : r == 17 ? 5 : ((r - 1) & 3) + 1) + (tid_ < 18 ? 0 : 6);
                    jl = txr * 4 + tyr;
                } else {
                    const int y = (tid_ - 36) & 15, right = tid_ >= 52;
                    jl = (right ? 12 : 0) + (y >> 2);
                    k = (right ? 16 : 12) + (y & 3);
                }
                const int gq = k / 5, i = k - gq * 5;
                const float *rp = red + (gq * 16 + jl) * H4_RED + 4 + i;
                const float s = ((rp[0] + rp[64 * H4_RED]) + rp[128 * H4_RED]) + rp[192 * H4_RED];
                a.ring[((size_t)tm * a.total_out + o0 + c2) * H4_RING + tid_] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same main loop as a plain 3x3 / stride 1 / pad 1 convolution (+ folded BN / bias, residual, ReLU, NHWC store) for the
// two shapes that keep V resident: cin == 64 with any number of 64-channel output tiles (ResNet layer 1: the output tiles take
// the place of the head's branches), or cout == 64 with cin a multiple of 64 (the CenterHead's shared 256 -> 64 layer: V is
// rebuilt per 64-channel chunk of the input, the 36 accumulators run through all chunks).  mmdet ResNet Bottleneck.conv2 /
// mmdet3d CenterHead.shared_conv as built by layers/backbones/lss_fpn.py:296-301 and layers/heads/bev_height_head.py:75-110.
struct F4ResArgs {
    const float *x, *u, *scale, *bias, *res;
    float *y;
    int x_ld, x_coff, y_ld, y_coff, res_ld, relu, h, w, wb_y, wb_x;
    int n_ct, n_kc;               // 64-channel output tiles, 64-channel input chunks (one of them is 1)
    unsigned u_bytes;
};

__global__ __launch_bounds__(256, 1) void conv_f4res_kernel(const F4ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x;
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * 16, ox0 = bx * 16;
    const float *const ximg = a.x + (size_t)img * a.h * a.w * a.x_ld + a.x_coff;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.u, 0, (int)a.u_bytes, 0x00020000);
    const unsigned w_lane = (unsigned)lane * 16u;
    unsigned w_off = (unsigned)wave * (unsigned)(a.n_ct * a.n_kc) * (unsigned)H4_BRANCH;
    f32x4 wf[H4_AHEAD];
#pragma unroll
    for (int r = 0; r < H4_AHEAD; ++r) wf[r] = h4_wload(rsrc, w_lane, w_off + (unsigned)(r * H4_FRAG));
    const int g = lane >> 4, j = lane & 15;

    // (Requesting the raw pixels of chunk kc + 1 before the main loop of chunk kc -- 144 registers held across it, or one
    // "touch" load per pixel -- does not pay: a wave's loads return in order, so the weight fragments behind them would wait
    // for HBM too, and the 144 registers spill.)
    for (int ct = 0; ct < a.n_ct; ++ct) {
        f32x4 acc[H4_POS];
        for (int kc = 0; kc < a.n_kc; ++kc) {
            if (ct == 0 || a.n_kc > 1) {       // (n_kc == 1: the one V serves every output tile)
                if (ct + kc > 0) H4_BARRIER();             // every wave is done with the previous chunk's V
                int tid_ = tid;
                asm volatile("" : "+v"(tid_));
                h4_build_v(smem, ximg + kc * 64, a.x_ld, a.h, a.w, oy0, ox0, tid_);
                H4_BARRIER();
            }
            if (kc == 0) h4_kquad<true>(acc, wf, smem + j * 16 + (g ^ j), rsrc, w_lane, w_off);
            else h4_kquad<false>(acc, wf, smem + j * 16 + (g ^ j), rsrc, w_lane, w_off);
#pragma unroll 1
            for (int q = 1; q < 4; ++q)
                h4_kquad<false>(acc, wf, smem + j * 16 + ((4 * q + g) ^ j), rsrc, w_lane, w_off + (unsigned)(q * H4_POS * H4_FRAG));
            w_off += (unsigned)H4_BRANCH;
        }
        // ---- epilogue: A^T M A per (tile, channel) inside the lane, then 4 channels x 16 pixels as 16-byte stores
        int tid_ = tid, oy0_ = oy0, ox0_ = ox0;
        asm volatile("" : "+v"(tid_), "+s"(oy0_), "+s"(ox0_));
        const int lane_ = tid_ & 63, g_ = lane_ >> 4, j_ = lane_ & 15;
        const int py0 = oy0_ + 4 * (j_ & 3), px0 = ox0_ + 4 * (j_ >> 2);
        const int co0 = ct * 64 + wave * 16 + 4 * g_;
        const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + co0) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + co0) : f32x4{0.f, 0.f, 0.f, 0.f};
        float hd[4][16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t[4][6];
#pragma unroll
            for (int jj = 0; jj < 6; ++jj)
                h4_at(acc[jj][r], acc[6 + jj][r], acc[12 + jj][r], acc[18 + jj][r], acc[24 + jj][r], acc[30 + jj][r], t[0][jj],
                      t[1][jj], t[2][jj], t[3][jj]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                h4_at(t[u][0], t[u][1], t[u][2], t[u][3], t[u][4], t[u][5], hd[r][u * 4], hd[r][u * 4 + 1], hd[r][u * 4 + 2],
                      hd[r][u * 4 + 3]);
        }
        const float floor_ = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int Y = py0 + u, X = px0 + v;
                if (Y < a.h && X < a.w) {
                    const size_t pix = ((size_t)img * a.h + Y) * a.w + X;
                    f32x4 o = f32x4{hd[0][u * 4 + v], hd[1][u * 4 + v], hd[2][u * 4 + v], hd[3][u * 4 + v]} * sc + sh;
                    if (a.res) o += *reinterpret_cast<const f32x4 *>(a.res + pix * a.res_ld + co0);
                    o = f32x4{fmaxf(o.x, floor_), fmaxf(o.y, floor_), fmaxf(o.z, floor_), fmaxf(o.w, floor_)};
                    *reinterpret_cast<f32x4 *>(a.y + pix * a.y_ld + a.y_coff + co0) = o;
                }
            }
    }
}

// U = G g G^T per (output channel, input channel) in the order the kernels stream it:
// [channel chunk 4][output tile (head: branch)][input chunk][k-quad 4][pos 36][lane 64][4]; lane = 16 g + m: output channel
// 64 tile + 16 chunk + m, input channel 64 input-chunk + 16 q + 4 g + (0..3); w: OIHW [n_ct * 64][cin_real][3][3]
__global__ void h4_pack_kernel(const float *__restrict__ w, int n_ct, int n_kc, int cin_real, float *__restrict__ dst, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long long r = i >> 8;
    const int p = (int)(r % 36);
    r /= 36;
    const int q = (int)(r & 3);
    r >>= 2;
    const int kc = (int)(r % n_kc);
    r /= n_kc;
    const int ct = (int)(r % n_ct), chunk = (int)(r / n_ct);
    const int co = ct * 64 + chunk * 16 + (lane & 15), ci = kc * 64 + 16 * q + 4 * (lane >> 4) + jj;
    double u = 0.0;
    if (ci < cin_real) {
        const float *gsrc = w + ((size_t)co * cin_real + ci) * 9;
        const double G[6][3] = {{0.25, 0, 0},           {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                                {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
        const int pi = p / 6, pj = p - pi * 6;
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x) u += G[pi][y] * (double)gsrc[y * 3 + x] * G[pj][x];
    }
    dst[i] = (float)u;
}

}  // namespace

extern "C" size_t sgv3d_centerhead_f4_weight_floats(int num_branches) {
    return num_branches > 0 ? (size_t)num_branches * 4 * (H4_BRANCH / 4) : 0;
}

// w1: OIHW [num_branches * 64, 64, 3, 3] f32 (the branches' first layers concatenated)
extern "C" int sgv3d_centerhead_f4_pack_weight(const float *w1, int num_branches, float *u_packed, void *stream) {
    SGV3D_REQUIRE(w1 && u_packed && num_branches > 0, "centerhead_f4_pack_weight: bad arguments");
    const long long total = (long long)sgv3d_centerhead_f4_weight_floats(num_branches);
    SGV3D_REQUIRE(total * 4 < 0xf0000000LL, "centerhead_f4_pack_weight: packed weights larger than 3.75 GiB");
    hipLaunchKernelGGL(h4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w1, num_branches, 1, 64, u_packed, total);
    return check_launch("h4_pack_kernel");
}

extern "C" int sgv3d_centerhead_branches_forward_f4(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                                    int num_branches, const float *u_packed, const float *scale1,
                                                    const float *bias1, int total_out, const float *w2, const float *bias2,
                                                    const int32_t *out_begin, float *out, void *workspace,
                                                    size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(x && u_packed && w2 && bias2 && out_begin && out, "centerhead_branches_forward_f4: null pointer");
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && num_branches > 0 && total_out > 0, "centerhead_branches_forward_f4: non-positive dimension");
    SGV3D_REQUIRE(cin == 64 && (x_ld & 3) == 0 && (x_coff & 3) == 0 && x_ld >= x_coff + cin,
                  "centerhead_branches_forward_f4: cin must be 64 (got %d), x_ld / x_coff multiples of 4", cin);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(u_packed) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(w2) & 15) == 0 && (!scale1 || (reinterpret_cast<uintptr_t>(scale1) & 15) == 0) &&
                      (!bias1 || (reinterpret_cast<uintptr_t>(bias1) & 15) == 0),
                  "centerhead_branches_forward_f4: x, u_packed, w2, scale1 and bias1 must be 16-B aligned");
    const size_t need = sgv3d_centerhead_branches_workspace_bytes(batch, h, w, total_out);
    if (!workspace || workspace_bytes < need)
        return fail(SGV3D_ENOSPACE, "centerhead_branches_forward_f4: workspace has %zu bytes, needs %zu", workspace_bytes, need);
    Head4Args a;
    a.x = x; a.u = u_packed; a.scale1 = scale1; a.bias1 = bias1; a.w2 = w2; a.bias2 = bias2; a.out_begin = out_begin;
    a.out = out; a.ring = static_cast<float *>(workspace);
    a.x_ld = x_ld; a.x_coff = x_coff; a.h = h; a.w = w; a.nb = num_branches; a.total_out = total_out;
    a.wb_y = cdiv(h, 16); a.wb_x = cdiv(w, 16);
    a.u_bytes = (unsigned)(sgv3d_centerhead_f4_weight_floats(num_branches) * 4);
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&head_wino4_kernel), (size_t)H4_LDS, lds_set))
        return fail(SGV3D_ELAUNCH, "centerhead_branches_forward_f4: cannot raise the dynamic LDS limit to %d", H4_LDS);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(head_wino4_kernel, dim3(batch * a.wb_y * a.wb_x), dim3(256), H4_LDS, st, a);
    const int rc = launch_head_ring_fixup(batch, h, w, total_out, a.ring, out, st);
    if (rc != SGV3D_OK) return rc;
    return check_launch("head_wino4_kernel");
}

// ---- plain convolution form (conv_f4res_kernel) ------------------------------------------------------------------------
extern "C" size_t sgv3d_conv3x3_f4res_weight_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout % 64 || cin % 64 || (cout != 64 && cin != 64)) return 0;
    return (size_t)(cout / 64) * (cin / 64) * 4 * (H4_BRANCH / 4);
}

// w: OIHW [cout, cin_real, 3, 3] f32, cin_real <= cin (the rest of the input channels multiply zeros)
extern "C" int sgv3d_conv3x3_f4res_pack_weight(const float *w, int cout, int cin_real, int cin, float *u_packed, void *stream) {
    const long long total = (long long)sgv3d_conv3x3_f4res_weight_floats(cout, cin);
    SGV3D_REQUIRE(w && u_packed && total > 0 && cin_real > 0 && cin_real <= cin,
                  "conv3x3_f4res_pack_weight: needs cout and cin multiples of 64, one of them 64 (got %d, %d / %d real)", cout, cin, cin_real);
    SGV3D_REQUIRE(total * 4 < 0xf0000000LL, "conv3x3_f4res_pack_weight: packed weights larger than 3.75 GiB");
    hipLaunchKernelGGL(h4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, cout / 64, cin / 64, cin_real, u_packed, total);
    return check_launch("h4_pack_kernel");
}

extern "C" int sgv3d_conv3x3_f4res_forward(const sgv3d_conv_desc *d, const float *x, const float *u_packed, const float *scale,
                                           const float *bias, const float *residual, float *y, void *stream) {
    SGV3D_REQUIRE(d && x && u_packed && y, "conv3x3_f4res_forward: null pointer");
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->mode == SGV3D_CONV_NORMAL,
                  "conv3x3_f4res_forward: only 3x3 / stride 1 / dilation 1 / pad 1, NHWC output (got k%dx%d s%d d%d p%d mode %d)", d->kh,
                  d->kw, d->stride, d->dil, d->pad, d->mode);
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->out_h == d->in_h && d->out_w == d->in_w,
                  "conv3x3_f4res_forward: bad map size");
    SGV3D_REQUIRE(sgv3d_conv3x3_f4res_weight_floats(d->cout, d->cin) > 0,
                  "conv3x3_f4res_forward: needs cin and cout multiples of 64, one of them 64 (got %d -> %d)", d->cin, d->cout);
    SGV3D_REQUIRE((d->x_ld & 3) == 0 && (d->x_coff & 3) == 0 && d->x_ld >= d->x_coff + d->cin && (d->y_ld & 3) == 0 &&
                      (d->y_coff & 3) == 0 && d->y_ld >= d->y_coff + d->cout && (!residual || ((d->res_ld & 3) == 0 && d->res_ld >= d->cout)),
                  "conv3x3_f4res_forward: channel strides / offsets must be multiples of 4 and cover the channels");
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u_packed) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual);
    SGV3D_REQUIRE((al & 15) == 0, "conv3x3_f4res_forward: pointers must be 16-B aligned");
    F4ResArgs a;
    a.x = x; a.u = u_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld; a.relu = d->relu;
    a.h = d->in_h; a.w = d->in_w; a.wb_y = cdiv(d->in_h, 16); a.wb_x = cdiv(d->in_w, 16);
    a.n_ct = d->cout / 64; a.n_kc = d->cin / 64;
    a.u_bytes = (unsigned)(sgv3d_conv3x3_f4res_weight_floats(d->cout, d->cin) * 4);
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_f4res_kernel), (size_t)H4_V_BYTES, lds_set))
        return fail(SGV3D_ELAUNCH, "conv3x3_f4res_forward: cannot raise the dynamic LDS limit to %d", H4_V_BYTES);
    hipLaunchKernelGGL(conv_f4res_kernel, dim3(d->batch * a.wb_y * a.wb_x), dim3(256), H4_V_BYTES, as_stream(stream), a);
    return check_launch("conv_f4res_kernel");
}
