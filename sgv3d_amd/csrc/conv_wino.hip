// Winograd F(2x2, 3x3) convolution for gfx950, fp32 in / fp32 accumulate, NHWC activations:
// the 3x3 / stride-1 / pad-1 convolutions of the path (ResNet 3x3s, HeightNet BasicBlocks, BEV trunk,
// CenterHead shared + fused branch layer) at 2.25x fewer MFMA flops than the implicit GEMM.  cuDNN
// picks the same algorithm family for these layers in the reference (layers/backbones/lss_fpn.py:
// 186-198,207-250; layers/heads/bev_height_head.py:75-110), so the rounding class is the reference's.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input tile d
//
// One kernel, nothing spilled to HBM in the Winograd domain:
//   * workgroup = 4 waves (2 x 2) = 64 tiles (8x8 tiles = 16x16 output pixels, or 4x16 tiles = 8x32 pixels)
//     x 64 output channels x all 16 Winograd positions; each wave owns 32 tiles x 32 channels x 16
//     positions = 16 MFMA accumulator tiles (256 accumulator registers: one wave per SIMD, the whole
//     512-register file).
//   * k-step = 8 input channels = 64 MFMAs per wave.  The raw input patch of the step (11.5 KB) goes
//     global -> registers -> LDS (two buffers, one barrier per step, in its middle); the pre-transformed
//     weights never touch LDS: packed [cout tile][k-step][16 pos][2][64 n][4] they ARE the MFMA B
//     fragments and are read by buffer_load_dwordx4 straight into a ring of fragment registers, three
//     position pairs ahead.  (An LDS-DMA version, global_load_lds_dwordx4 for both operands, was slower:
//     a piece costs the issuing wave 60-180 cycles.)
//   * the input transform B^T d B runs in registers on the fly (each lane reads the 16 raw float4 of its
//     tile once per step: 64 add/sub per component), feeding 16 x 4 MFMAs per step and wave.
//   * the patch image in LDS is conflict-free: [k/4][row][column parity][column/2 (padded)] so the 16
//     lanes of a ds_read_b128 group (8 tiles x 2 tile rows, or 16 tiles of a row) hit 16 distinct slots.
//   * at one wave per SIMD VALU instructions and fp32 MFMAs do not overlap (tools/ubench/mfma_rate.hip):
//     the MFMAs of two positions alternate and everything else is gathered into two MFMA gaps per pair.
//   * the output transform A^T M A is lane-local (the 16 positions of one (tile, channel) element sit in
//     the same lane / register index of the 16 accumulator tiles), followed by a slim epilogue.
//   * split-K over input-channel steps (grid.y) for layers with too few tiles to fill 256 CUs; partial
//     OUTPUT tiles (the transform is linear) go to the workspace and the igemm reduce kernel finishes.
// Variants: conv_wino_half_kernel (64 tiles x 32 channels, the 16 positions split over wave pairs: two workgroups per CU --
// what the pipeline with several frames in flight prefers), conv_wino_resident_kernel (patch of all k-steps resident in
// LDS, walks over the cout tiles) and conv_wino_head_kernel (both CenterHead branch layers, hidden maps never leave LDS).
#include <type_traits>
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WK = 8;                           // input channels per k-step
// Shape of the 64 tiles of a workgroup: TR x TC tiles (8 x 8 = 16x16 output pixels, or 4 x 16 = 8x32 pixels
// for feature maps like 54x96 whose height is far from a multiple of 16).  Patch = (2TR+2) x (2TC+2) pixels,
// stored [channel half][row][column parity][column / 2 (TC+1 used, padded to HALF)]; HALF makes the 16
// lanes of a ds_read_b128 group (8 tiles x 2 tile rows, or 16 tiles of one row) hit 16 distinct 16-B slots.
template <int TC>
struct WinoGeom {
    static constexpr int TR = 64 / TC;
    static constexpr int RW = 32 / TC;                 // tile rows per wave
    static constexpr int ROWS = 2 * TR + 2;            // 18 | 10
    static constexpr int HALF = TC == 8 ? 10 : 18;     // 9 | 17 used
    static constexpr int PLANE = ROWS * 2 * HALF;      // 360 slots per 4-channel plane (both shapes)
    static constexpr int USED = 2 * PLANE;             // 720
    static_assert(PLANE == 360, "the patch must fit 3 slots per thread");
};
constexpr int A_SLOTS = 768;                    // 3 slots per thread
constexpr int W_STEP = 16 * 2 * 64 * 4;         // floats of one k-step of one 64-channel tile (32 KB)
constexpr int W_POS = 2 * 64 * 4;               // floats per Winograd position inside a step
constexpr int kWinoLds = 2 * A_SLOTS * 16;      // two patch buffers (24 KB)

#define SGV3D_SB() __builtin_amdgcn_sched_barrier(0)
// every wave's patch writes have landed (lgkmcnt) -> barrier.  Outstanding global loads (weight
// fragments, the next patch) stay in flight: unlike __syncthreads() this does not wait on vmcnt.
#define SGV3D_WINO_PUBLISH() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// B^T rows of F(2x2,3x3): (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
template <int R>
__device__ __forceinline__ f32x4 wino_bt(const f32x4 &x0, const f32x4 &x1, const f32x4 &x2, const f32x4 &x3) {
    if constexpr (R == 0) return x0 - x2;
    else if constexpr (R == 1) return x1 + x2;
    else if constexpr (R == 2) return x2 - x1;
    else return x1 - x3;
}

// Per-wave state of the two operand streams.  Both are read with buffer loads: the resource and the
// step / position part of the address live in scalar registers, the lane part is one constant VGPR, so
// a fragment load costs no vector ALU work (on this chip VALU instructions and fp32 MFMAs of one wave do
// not overlap), and out-of-image patch lanes simply carry an out-of-range offset (buffer loads return 0).
struct WinoStreams {
    __amdgpu_buffer_rsrc_t w_rsrc, x_rsrc;
    unsigned w_cur, w_next;        // byte offset of the current / next k-step of this cout tile (uniform)
    unsigned w_lane;               // this lane's byte offset inside a position: (h*64 + wn*32 + t) * 16
    unsigned x_step;               // byte offset of the k-step whose patch is fetched next (uniform)
    unsigned x0, x1, x2;           // this thread's three patch slots: byte offset of the pixel (or out of range)
    f32x4 stage0, stage1, stage2;  // patch of the next step on its way global -> registers -> LDS
    f32x4 *a_wr;                   // LDS slot 0 of this thread in the buffer the next patch is written to
};

__device__ __forceinline__ f32x4 wino_buffer_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// Two Winograd positions P, P+1 of the current step: 2 x 4 MFMAs (k = 4h + 0..3) alternating between the
// two accumulators.  Everything else the wave has to issue is gathered into two of the eight MFMA gaps
// (measured on gfx950, one wave per SIMD: a gap that carries other instructions costs ~10 cycles plus ~3
// per VALU instruction, because fp32 MFMAs and VALU share the SIMD's fp32 lanes -- so fewer, fuller gaps):
//   gap 1 (memory): the weight fragments three pairs ahead (P+6, P+7; positions >= 16 are the next
//     step's) -- buffer_load_dwordx4 straight into the fragment registers: the packed layout IS the
//     fragment layout, each half-wave reads 512 contiguous bytes, the two waves sharing a channel half
//     hit in L1; ring of 8 fragments, ~1500 cycles of prefetch distance.  Plus the patch pipeline's
//     memory operations: pairs 0..4 write the staged patch of step s+1 to LDS (published by the barrier in
//     the middle of the step), pair 8 reads it back as 4x4 tiles, pairs 10..14 fetch the patch of step s+2.
//   gap 2 (VALU): the row pass (V = T B) of the next pair, and in pairs 10..14 the column pass
//     T = B^T d of the next step's patch, row by row into tc as soon as the current step's row is dead
//     (rows 0, 1 in pair 10, row 2 in pair 12, row 3 in pair 14).
#define SGV3D_WINO_MFMA(P, C, V, BF) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.C, BF.C, acc[P], 0, 0, 0)
// ZERO: the pair's accumulators start from zero HERE (first k-step of an output tile): its first MFMA takes the constant 0
// as C instead of the register, so the 2 x 16 accumulator registers need no clearing (256 v_accvgpr_write per output tile
// otherwise -- vector instructions that cost MFMA time; the fused head starts 36 tiles per workgroup).
template <int P, int TC, bool RESIDENT = false, int DIST = 3, int HALF = WinoGeom<TC>::HALF, bool ZERO = false>
__device__ __forceinline__ void wino_pair(f32x16 (&acc)[16], f32x4 (&tc)[4][4], f32x4 (&raw)[4][4], f32x4 &vc0,
                                          f32x4 &vc1, f32x4 &vn0, f32x4 &vn1, f32x4 (&wf)[8], WinoStreams &st,
                                          const f32x4 *An) {
    static_assert((P & 1) == 0, "pairs start at even positions");
    constexpr int R0 = P & 7, R1 = (P + 1) & 7;          // ring slots of this pair
    static_assert(DIST == 3, "fragment ring: three pairs ahead");
    constexpr int L0 = (P + 6) & 7, L1 = (P + 7) & 7;    // ring slots (= those of pair P-2) refilled now
    const f32x4 b0 = wf[R0], b1 = wf[R1];
    if constexpr (ZERO) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc0.x, b0.x, z, 0, 0, 0);
    } else {
        SGV3D_WINO_MFMA(P, x, vc0, b0);
    }
    SGV3D_SB();
    // ---- gap 1: memory ----
    if constexpr (P + 6 < 16) wf[L0] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_cur + (P + 6) * (W_POS * 4));
    else wf[L0] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_next + (P + 6 - 16) * (W_POS * 4));
    if constexpr (P + 7 < 16) wf[L1] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_cur + (P + 7) * (W_POS * 4));
    else wf[L1] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_next + (P + 7 - 16) * (W_POS * 4));
    if constexpr (!RESIDENT && (P == 0 || P == 2 || P == 4)) {
        st.a_wr[(P / 2) * 256] = P == 0 ? st.stage0 : P == 2 ? st.stage1 : st.stage2;   // slot (P/2)*256 + tid
    } else if constexpr (P == 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                raw[i][j] = An[i * (2 * HALF) + (j & 1) * HALF + (j >> 1)];
    } else if constexpr (!RESIDENT && P == 10) {
        st.stage0 = wino_buffer_load(st.x_rsrc, st.x0, st.x_step);
    } else if constexpr (!RESIDENT && P == 12) {
        st.stage1 = wino_buffer_load(st.x_rsrc, st.x1, st.x_step);
    } else if constexpr (!RESIDENT && P == 14) {
        st.stage2 = wino_buffer_load(st.x_rsrc, st.x2, st.x_step);
    }
    SGV3D_SB();
    if constexpr (ZERO) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[P + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc1.x, b1.x, z, 0, 0, 0);
    } else {
        SGV3D_WINO_MFMA(P + 1, x, vc1, b1);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, y, vc0, b0);
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, y, vc1, b1);
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, z, vc0, b0);
    SGV3D_SB();
    // ---- gap 2: vector ALU ----
    {
        constexpr int Q0 = (P + 2) & 15, Q1 = (P + 3) & 15;
        vn0 = wino_bt<(Q0 & 3)>(tc[Q0 >> 2][0], tc[Q0 >> 2][1], tc[Q0 >> 2][2], tc[Q0 >> 2][3]);
        vn1 = wino_bt<(Q1 & 3)>(tc[Q1 >> 2][0], tc[Q1 >> 2][1], tc[Q1 >> 2][2], tc[Q1 >> 2][3]);
    }
    if constexpr (P == 10) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tc[0][j] = wino_bt<0>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[1][j] = wino_bt<1>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
        }
    } else if constexpr (P == 12) {
#pragma unroll
        for (int j = 0; j < 4; ++j) tc[2][j] = wino_bt<2>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, z, vc1, b1);
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, w, vc0, b0);
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, w, vc1, b1);
    SGV3D_SB();
    if constexpr (P == 14) {   // row 3 of the current step died with vn1 (position 15) above
#pragma unroll
        for (int j = 0; j < 4; ++j) tc[3][j] = wino_bt<3>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
        SGV3D_SB();
    }
}

// Output transform Y = A^T M A of the 16 tiles a lane holds (one output channel each) and their stores;
// see the epilogue comment in the kernel for the addressing.  PARTIAL: raw split-K partials.
template <int TC, bool PARTIAL, bool HAS_RES>
__device__ __forceinline__ void wino_store(f32x16 (&acc)[16], __amdgpu_buffer_rsrc_t y_rsrc, __amdgpu_buffer_rsrc_t r_rsrc,
                                           const unsigned (&voff)[TC / 2][2], const unsigned (&roff)[TC / 2][2], unsigned rowpitch,
                                           unsigned rpitch, int oy_wave, int out_h, float sc, float sh, float gt,
                                           float floor_) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        // accumulator element e is tile tw = 8*(e>>2) + 4h + (e&3) of the wave: row tw / TC, column tw % TC
        float r0[4], r1[4];   // rows of A^T M:  r0 = m0 + m1 + m2,  r1 = m1 - m2 - m3   (per Winograd column j)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[j][e], m1 = acc[4 + j][e], m2 = acc[8 + j][e], m3 = acc[12 + j][e];
            r0[j] = m0 + m1 + m2;
            r1[j] = m1 - m2 - m3;
        }
        const float yv[2][2] = {{r0[0] + r0[1] + r0[2], r0[1] - r0[2] - r0[3]},
                                {r1[0] + r1[1] + r1[2], r1[1] - r1[2] - r1[3]}};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = 2 * ((8 * (e >> 2)) / TC) + dy;
            const int xi = ((e >> 2) % (TC / 8)) * 4 + (e & 3);
            if (oy_wave + yy < out_h) {   // wave-uniform
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v = yv[dy][dx];
                    if constexpr (!PARTIAL) {
                        v = v * sc + sh;
                        if constexpr (HAS_RES)
                            v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, roff[xi][dx], yy * rpitch, 0));
                        v = fmaxf(v, floor_) * gt;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, voff[xi][dx], yy * rowpitch, 0);
                }
            }
        }
    }
}

// Epilogue of one (block, cout tile): output transform, folded BN / bias, residual, ReLU, SE gate, store.
// At one wave per SIMD nothing hides it, so it is kept to a few instructions per output: a lane holds ONE
// output channel (col) of 16 tiles x 2x2 pixels, every channel-only term (scale, shift, gate, channel
// part of the address) is hoisted, and the stores are buffer stores whose address is
//   [uniform base in the resource] + [one of 8 per-lane VGPR offsets: lane's channel, its wave's tile
//   rows, column 2c+dx -- or out of range, which drops the store, when that column is outside the image]
//   + [scalar offset of tile row / dy].
// Same arithmetic as conv_epilogue_store (conv_common.hpp).
template <int TC>
__device__ __forceinline__ void wino_epilogue(const ConvArgs &a, f32x16 (&acc)[16], int tn, int img, int oy0, int ox0,
                                              int wm, int wn, int h, int t) {
    // Opaque copies: everything derived from the lane / block coordinates below (address tables, bounds) is
    // computed HERE, after the main loop, instead of being hoisted in front of it and kept -- or spilled --
    // across a loop that has no registers to spare.
    asm volatile("" : "+v"(h), "+v"(t), "+s"(wm), "+s"(wn), "+s"(oy0), "+s"(ox0), "+s"(tn), "+s"(img));
    // At one wave per SIMD nothing hides the epilogue, so it is kept to a few instructions per output:
    // a lane holds ONE output channel (col) of 16 tiles x 2x2 pixels, every channel-only term (scale,
    // shift, gate, channel part of the address) is hoisted, and the stores are buffer stores whose
    // address is  [uniform base in the resource]  +  [one of 8 per-lane VGPR offsets: lane's channel, its
    // wave's tile rows, column 2c+dx -- or out of range, which drops the store, when that column is
    // outside the image]  +  [scalar offset of tile row / dy].  Same arithmetic as conv_epilogue_store.
    const int col = tn * 64 + wn * 32 + t;
    const bool partial = a.split_k > 1;
    const long long row0 = ((long long)img * a.out_h + oy0) * a.out_w + ox0;   // first pixel of the block
    const char *ybase;         // uniform
    unsigned pixstride;        // bytes between horizontally adjacent output pixels
    unsigned lane_off;         // this lane's channel
    float sc = 1.f, sh = 0.f, gt = 1.f, floor_ = -__builtin_inff();
    if (partial) {
        ybase = reinterpret_cast<const char *>(a.ws + ((size_t)blockIdx.y * a.M + row0) * a.N + tn * 64);
        pixstride = a.N * 4u;
        lane_off = (wn * 32 + t) * 4u;
    } else {
        if (col < a.N) {
            if (a.scale) sc = a.scale[col];
            if (a.bias) sh = a.bias[col];
            if (a.gate) gt = a.gate[(size_t)img * a.cout + col];
        }
        if (a.relu) floor_ = 0.f;
        if (a.mode == SGV3D_CONV_NORMAL) {
            ybase = reinterpret_cast<const char *>(a.y + row0 * a.y_ld + a.y_coff + tn * 64);
            pixstride = a.y_ld * 4u;
            lane_off = (wn * 32 + t) * 4u;
        } else if (a.mode == SGV3D_CONV_NCHW_OUT) {
            const long long hw = (long long)a.out_h * a.out_w;
            ybase = reinterpret_cast<const char *>(a.y + ((size_t)img * a.y_ld + a.y_coff + tn * 64) * hw +
                                                   (long long)oy0 * a.out_w + ox0);
            pixstride = 4u;
            lane_off = (unsigned)((wn * 32 + t) * hw * 4);
        } else {  // GROUP_PLANES: [cout/g][M][g], g = a.ks
            const int grp0 = (tn * 64) / a.ks, grp = col / a.ks;
            ybase = reinterpret_cast<const char *>(a.y + ((size_t)grp0 * a.M + row0) * a.ks);
            pixstride = a.ks * 4u;
            lane_off = (unsigned)((((size_t)(grp - grp0) * a.M) * a.ks + (col - grp * a.ks)) * 4);
        }
    }
    const unsigned rowpitch = a.out_w * pixstride;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ybase, 0, (int)0xffffff00u, 0x00020000);
    const bool has_res = !partial && a.res != nullptr;
    const unsigned rpix = a.res_ld * 4u, rpitch = a.out_w * rpix;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_res ? a.res + row0 * a.res_ld + tn * 64 : a.zeros), 0, has_res ? (int)0xffffff00u : 0, 0x00020000);
    constexpr int RW2 = 2 * WinoGeom<TC>::RW;      // output rows per wave
    unsigned voff[TC / 2][2], roff[TC / 2][2];
#pragma unroll
    for (int c = 0; c < TC / 2; ++c)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = 8 * h + 2 * ((c >> 2) * 8 + (c & 3)) + dx;   // column of tile 8*(c>>2) + 4h + (c&3)
            const bool ok = (col < a.N) & (ox0 + xx < a.out_w);
            voff[c][dx] = ok ? lane_off + (wm * RW2) * rowpitch + xx * pixstride : 0xffffffffu;
            roff[c][dx] = ok ? (wn * 32 + t) * 4u + (wm * RW2) * rpitch + xx * rpix : 0xffffffffu;
        }
    const int oy_wave = oy0 + wm * RW2;
    if (partial) wino_store<TC, true, false>(acc, y_rsrc, r_rsrc, voff, roff, rowpitch, rpitch, oy_wave, a.out_h, sc, sh, gt, floor_);
    else if (has_res) wino_store<TC, false, true>(acc, y_rsrc, r_rsrc, voff, roff, rowpitch, rpitch, oy_wave, a.out_h, sc, sh, gt, floor_);
    else wino_store<TC, false, false>(acc, y_rsrc, r_rsrc, voff, roff, rowpitch, rpitch, oy_wave, a.out_h, sc, sh, gt, floor_);
}

template <int TC>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a) {
    using G = WinoGeom<TC>;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];

    // ---- XCD-aware tile mapping (same bijection as the implicit-GEMM kernel) ----------------------
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tn = logical / a.tiles_m;
    const int tm = logical - tn * a.tiles_m;
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * (2 * G::TR), ox0 = bx * (2 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, t = lane & 31;

    const int nsteps_all = a.cin / WK;
    const int kb = (int)((long long)nsteps_all * blockIdx.y / a.split_k);
    const int ke = (int)((long long)nsteps_all * (blockIdx.y + 1) / a.split_k);

    // ---- the two operand streams -------------------------------------------------------------------
    // Patch: thread tid owns LDS slots tid, tid + 256, tid + 512 of each buffer; slot -> (channel half,
    // row, column parity, column / 2) -> one input pixel, or an out-of-range offset (reads as zeros) for
    // the conv padding, ragged image edges and the layout's pad slots.
    WinoStreams st;
    const unsigned x_bytes = (unsigned)((size_t)a.M * a.x_ld * sizeof(float));   // in == out size (host checks < 4 GiB)
    st.x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)x_bytes, 0x00020000);
    unsigned xoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 256 + tid;
        const int hh = s / G::PLANE, rem = s - hh * G::PLANE;
        const int row = rem / (2 * G::HALF), r2 = rem - row * (2 * G::HALF);
        const int par = r2 / G::HALF, ch = r2 - par * G::HALF;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + 2 * ch + par;
        const bool ok = (s < G::USED) & (ch < TC + 1) & (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
        xoff[i] = ok ? (unsigned)((((size_t)(img * a.in_h + iy) * a.in_w + ix) * a.x_ld + a.x_coff + hh * 4) * sizeof(float))
                     : 0xfffffff0u - (unsigned)(a.cin * sizeof(float));   // stays out of range for every k-step
    }
    st.x0 = xoff[0];
    st.x1 = xoff[1];
    st.x2 = xoff[2];
    st.x_step = (unsigned)(kb * WK * sizeof(float));
    // Weights: packed [cout tile][k-step][pos][channel half][64 n][4] = exactly the MFMA B fragments.
    const unsigned w_bytes = (unsigned)((size_t)a.tiles_n * nsteps_all * W_STEP * sizeof(float));
    st.w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)w_bytes, 0x00020000);
    st.w_cur = (unsigned)(((size_t)tn * nsteps_all + kb) * W_STEP * sizeof(float));
    st.w_next = kb + 1 < ke ? st.w_cur + W_STEP * 4 : st.w_cur;
    st.w_lane = (unsigned)(h * 64 + wn * 32 + t) * 16u;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    const int abase = (h * G::ROWS + 2 * (wm * G::RW + t / TC)) * (2 * G::HALF) + t % TC;

    // ---- prologue: patch of step kb through LDS, transformed; fragments of positions 0..5; patch of
    //      step kb+1 on its way ------------------------------------------------------------------------
    f32x4 tc[4][4], raw[4][4], wf[8], va0, va1, vb0, vb1;
    st.stage0 = wino_buffer_load(st.x_rsrc, st.x0, st.x_step);
    st.stage1 = wino_buffer_load(st.x_rsrc, st.x1, st.x_step);
    st.stage2 = wino_buffer_load(st.x_rsrc, st.x2, st.x_step);
#pragma unroll
    for (int p = 0; p < 6; ++p) wf[p] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_cur + p * (W_POS * 4));
    smem[tid] = st.stage0;
    smem[256 + tid] = st.stage1;
    smem[512 + tid] = st.stage2;
    if (kb + 1 < ke) st.x_step += WK * sizeof(float);
    st.stage0 = wino_buffer_load(st.x_rsrc, st.x0, st.x_step);
    st.stage1 = wino_buffer_load(st.x_rsrc, st.x1, st.x_step);
    st.stage2 = wino_buffer_load(st.x_rsrc, st.x2, st.x_step);
    SGV3D_WINO_PUBLISH();
    {
        const f32x4 *const A = smem + abase;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = A[i * (2 * G::HALF) + (j & 1) * G::HALF + (j >> 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tc[0][j] = wino_bt<0>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[1][j] = wino_bt<1>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[2][j] = wino_bt<2>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[3][j] = wino_bt<3>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
        }
        va0 = wino_bt<0>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
        va1 = wino_bt<1>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
    }

    // ---- main loop: one barrier per step, in its middle ---------------------------------------------
    // Patch buffers alternate: step s+1's patch is written to buffer (s+1)&1 in the first half of step
    // s (its previous content, patch s-1, was last read in the second half of step s-2, and every wave
    // has passed the barrier of step s-1 since), published by the barrier of step s, read back in the
    // second half of step s.  In the last step the patch / fragment prefetches re-read the last valid
    // step (offsets stop advancing) and their results are dropped.
    for (int s = kb; s < ke; ++s) {
        const int nb = (s + 1 - kb) & 1;
        st.a_wr = smem + nb * A_SLOTS + tid;
        const f32x4 *const An = smem + nb * A_SLOTS + abase;
        wino_pair<0, TC>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
        wino_pair<2, TC>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
        wino_pair<4, TC>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
        wino_pair<6, TC>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
        SGV3D_WINO_PUBLISH();
        if (s + 2 < ke) st.x_step += WK * sizeof(float);   // the patch fetched in the second half is step s+2's
        SGV3D_SB();
        wino_pair<8, TC>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
        wino_pair<10, TC>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
        wino_pair<12, TC>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
        wino_pair<14, TC>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
        st.w_cur = st.w_next;
        if (s + 2 < ke) st.w_next += W_STEP * 4;
    }

    wino_epilogue<TC>(a, acc, tn, img, oy0, ox0, wm, wn, h, t);
}

// ------------------------------------------------------------------------------------------------
// "Half-position" variant: two workgroups per CU.
// conv_wino_kernel owns the whole register file (256 accumulator registers per wave, one wave per SIMD): whenever its
// wave is not issuing an MFMA -- the input transform, the barrier in the middle of a step, prologue, epilogue -- the
// SIMD's matrix pipe idles, and no other workgroup can share the CU (MFMA busy 39-48 % over a launch).  Here a workgroup
// is 64 tiles x 32 output channels and its four waves are (tile half) x (Winograd rows i in {0,1} | {2,3}): a wave holds
// 8 of the 16 position accumulators (128 registers), needs three of the four raw tile rows (T0 = d0 - d2, T1 = d1 + d2
// | T2 = d2 - d1, T3 = d1 - d3) and half the weight fragments, so per-MFMA transform work and operand traffic are those of
// the full kernel -- but at <= 256 registers two workgroups share a CU and fill each other's gaps, there are twice as many
// (half as long) workgroups to balance, and the split-K the full kernel needs to fill 256 CUs is rarely necessary.
// The output transform Y = A^T M A is linear in the positions: each wave transforms its two rows of M, keeps the output
// row it owns (rows {0,1} -> dy = 0, rows {2,3} -> dy = 1), hands the other row's partial sums to its partner wave through
// LDS (8 KB per wave), adds what it receives and runs the common epilogue on half of the pixels.
// Weights: the layout of conv_wino_kernel ([cout tile of 64][k-step][16 pos][2][64 n][4]); the workgroup's 32 channels
// are one half of a tile.  Schedule: plain -- one position at a time (V = row pass on the fly, 4 MFMAs, refill of the
// position's fragment for the next step); the co-resident workgroup covers the latencies this leaves open.
// ------------------------------------------------------------------------------------------------
constexpr int kWinoHalfLds = 2 * A_SLOTS * 16 > 4 * 16 * 64 * 8 ? 2 * A_SLOTS * 16 : 4 * 16 * 64 * 8;   // patch buffers | exchange

template <int TC>
__global__ __launch_bounds__(256, 2) void conv_wino_half_kernel(const ConvArgs a) {
    using G = WinoGeom<TC>;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];

    // ---- tile mapping: (block of 64 tiles) x (32 output channels); a.tiles_n counts 32-channel tiles here ----
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tn2 = (int)((unsigned)logical / (unsigned)a.tiles_m);
    const int tm = logical - tn2 * a.tiles_m;
    const int tn = tn2 >> 1, wn = tn2 & 1;          // 64-channel tile of the packed weights, half inside it
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * (2 * G::TR), ox0 = bx * (2 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, ph = wave & 1;        // tile half, position half (Winograd rows 2 ph, 2 ph + 1)
    const int h = lane >> 5, t = lane & 31;

    const int nsteps_all = a.cin / WK;
    int kb = 0, ke = nsteps_all;
    if (a.split_k > 1) {
        kb = (int)((unsigned)nsteps_all * blockIdx.y / (unsigned)a.split_k);
        ke = (int)((unsigned)nsteps_all * (blockIdx.y + 1) / (unsigned)a.split_k);
    }

    // ---- operand streams (as in conv_wino_kernel) ------------------------------------------------------
    const unsigned x_bytes = (unsigned)((size_t)a.M * a.x_ld * sizeof(float));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)x_bytes, 0x00020000);
    unsigned xoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 256 + tid;
        const int hh = s / G::PLANE, rem = s - hh * G::PLANE;
        const int row = rem / (2 * G::HALF), r2 = rem - row * (2 * G::HALF);
        const int par = r2 / G::HALF, ch = r2 - par * G::HALF;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + 2 * ch + par;
        const bool ok = (s < G::USED) & (ch < TC + 1) & (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
        xoff[i] = ok ? (unsigned)((((size_t)(img * a.in_h + iy) * a.in_w + ix) * a.x_ld + a.x_coff + hh * 4) * sizeof(float))
                     : 0xfffffff0u - (unsigned)(a.cin * sizeof(float));
    }
    unsigned x_step = (unsigned)(kb * WK * sizeof(float));
    const unsigned w_bytes = (unsigned)((size_t)((a.tiles_n + 1) >> 1) * nsteps_all * W_STEP * sizeof(float));
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)w_bytes, 0x00020000);
    // this wave's eight positions start at position 8 ph of the step
    unsigned w_cur = (unsigned)((((size_t)tn * nsteps_all + kb) * W_STEP + (size_t)ph * 8 * W_POS) * sizeof(float));
    unsigned w_next = kb + 1 < ke ? w_cur + W_STEP * 4 : w_cur;
    const unsigned w_lane = (unsigned)(h * 64 + wn * 32 + t) * 16u;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // raw tile row R (0..3) / column j of this lane's tile in a patch buffer
    const int abase = (h * G::ROWS + 2 * (wm * G::RW + t / TC)) * (2 * G::HALF) + t % TC;
#define SGV3D_WH_RAW(A, R, J) (A)[(R) * (2 * G::HALF) + ((J) & 1) * G::HALF + ((J) >> 1)]
    // one transformed row of one column from the patch buffer A:  ROW 0: ph ? d2 - d1 : d0 - d2,  ROW 1: ph ? d1 - d3 : d1 + d2
#define SGV3D_WH_T(A, ROW, J)                                                                         \
    ((ROW) == 0 ? (ph ? SGV3D_WH_RAW(A, 2, J) - SGV3D_WH_RAW(A, 1, J) : SGV3D_WH_RAW(A, 0, J) - SGV3D_WH_RAW(A, 2, J)) \
                : (ph ? SGV3D_WH_RAW(A, 1, J) - SGV3D_WH_RAW(A, 3, J) : SGV3D_WH_RAW(A, 1, J) + SGV3D_WH_RAW(A, 2, J)))

    // ---- prologue -----------------------------------------------------------------------------------
    f32x4 tc[2][4], wf[8], stage0, stage1, stage2, va0, va1, vb0, vb1;
    stage0 = wino_buffer_load(x_rsrc, xoff[0], x_step);
    stage1 = wino_buffer_load(x_rsrc, xoff[1], x_step);
    stage2 = wino_buffer_load(x_rsrc, xoff[2], x_step);
#pragma unroll
    for (int p = 0; p < 8; ++p) wf[p] = wino_buffer_load(w_rsrc, w_lane, w_cur + p * (W_POS * 4));
    smem[tid] = stage0;
    smem[256 + tid] = stage1;
    smem[512 + tid] = stage2;
    if (kb + 1 < ke) x_step += WK * sizeof(float);
    stage0 = wino_buffer_load(x_rsrc, xoff[0], x_step);
    stage1 = wino_buffer_load(x_rsrc, xoff[1], x_step);
    stage2 = wino_buffer_load(x_rsrc, xoff[2], x_step);
    SGV3D_WINO_PUBLISH();
    {
        const f32x4 *const A = smem + abase;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tc[0][j] = SGV3D_WH_T(A, 0, j);
            tc[1][j] = SGV3D_WH_T(A, 1, j);
        }
        va0 = wino_bt<0>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
        va1 = wino_bt<1>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
    }

    // ---- main loop: one barrier per step, after the second pair --------------------------------------
    // A step = 8 positions = 4 pairs; the MFMAs of a pair alternate between its two accumulators and everything else is pinned
    // into two gaps per pair (the scheduler otherwise issues the four MFMAs of an accumulator back to back -- a dependent
    // chain -- and sinks the loads to their first use):
    //   gap 1 (memory): refill of the previous pair's two weight fragments with the next step's (a full step of prefetch
    //     distance); the patch pipeline -- pair 0 writes the staged patch of step s+1 to LDS, the barrier after pair 2 publishes
    //     it, pair 4 fetches the patch of step s+2; the raw tile columns the row pass of this gap's pair needs.
    //   gap 2 (VALU): the column pass (V = T B) of the next pair, and the row pass T = B^T d of the next step, row by row as
    //     soon as the current row is dead: row 0 (positions 0..3, done after pair 2) in pairs 4 / 6 from the buffer just
    //     published, row 1 (positions 4..7) in pairs 0 / 2 of the next step.
    // Patch buffers: step s+1's patch goes to buffer (s+1)&1 at pair 0 of step s (its previous content, patch s-1, was last
    // read in pair 2 of step s-1 and every wave has passed that step's barrier since), and is read from pair 4 of step s to
    // pair 2 of step s+1.  In the last step the prefetches re-read the last valid step and their results are dropped.
#define SGV3D_WH_MFMA(P, C, V, BF) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.C, BF.C, acc[P], 0, 0, 0)
    auto pair = [&](auto Pc, f32x4 &vc0, f32x4 &vc1, f32x4 &vn0, f32x4 &vn1, const f32x4 *Acur, const f32x4 *Anew,
                    f32x4 *a_wr) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int L0 = (P + 6) & 7, L1 = (P + 7) & 7;        // the previous pair's fragments, dead now
        const f32x4 b0 = wf[P], b1 = wf[P + 1];
        SGV3D_WH_MFMA(P, x, vc0, b0);
        SGV3D_SB();
        // ---- gap 1: memory ----
        wf[L0] = wino_buffer_load(w_rsrc, w_lane, (P == 0 ? w_cur : w_next) + L0 * (W_POS * 4));
        wf[L1] = wino_buffer_load(w_rsrc, w_lane, (P == 0 ? w_cur : w_next) + L1 * (W_POS * 4));
        if constexpr (P == 0) {
            a_wr[0] = stage0;
            a_wr[256] = stage1;
            a_wr[512] = stage2;
        } else if constexpr (P == 4) {
            stage0 = wino_buffer_load(x_rsrc, xoff[0], x_step);
            stage1 = wino_buffer_load(x_rsrc, xoff[1], x_step);
            stage2 = wino_buffer_load(x_rsrc, xoff[2], x_step);
        }
        f32x4 t0, t1;
        if constexpr (P == 0) { t0 = SGV3D_WH_T(Acur, 1, 0); t1 = SGV3D_WH_T(Acur, 1, 1); }
        else if constexpr (P == 2) { t0 = SGV3D_WH_T(Acur, 1, 2); t1 = SGV3D_WH_T(Acur, 1, 3); }
        else if constexpr (P == 4) { t0 = SGV3D_WH_T(Anew, 0, 0); t1 = SGV3D_WH_T(Anew, 0, 1); }
        else { t0 = SGV3D_WH_T(Anew, 0, 2); t1 = SGV3D_WH_T(Anew, 0, 3); }
        SGV3D_SB();
        SGV3D_WH_MFMA(P + 1, x, vc1, b1);
        SGV3D_SB();
        SGV3D_WH_MFMA(P, y, vc0, b0);
        SGV3D_SB();
        SGV3D_WH_MFMA(P + 1, y, vc1, b1);
        SGV3D_SB();
        SGV3D_WH_MFMA(P, z, vc0, b0);
        SGV3D_SB();
        // ---- gap 2: vector ALU ----
        if constexpr (P == 0) { tc[1][0] = t0; tc[1][1] = t1; }
        else if constexpr (P == 2) { tc[1][2] = t0; tc[1][3] = t1; }
        else if constexpr (P == 4) { tc[0][0] = t0; tc[0][1] = t1; }      // (row 0 of this step was last read in gap 2 of pair 0)
        else { tc[0][2] = t0; tc[0][3] = t1; }
        {
            constexpr int Q = (P + 2) & 7;                       // next pair: positions Q, Q + 1 of row Q >> 2
            vn0 = wino_bt<(Q & 3)>(tc[Q >> 2][0], tc[Q >> 2][1], tc[Q >> 2][2], tc[Q >> 2][3]);
            vn1 = wino_bt<((Q + 1) & 3)>(tc[Q >> 2][0], tc[Q >> 2][1], tc[Q >> 2][2], tc[Q >> 2][3]);
        }
        SGV3D_SB();
        SGV3D_WH_MFMA(P + 1, z, vc1, b1);
        SGV3D_SB();
        SGV3D_WH_MFMA(P, w, vc0, b0);
        SGV3D_SB();
        SGV3D_WH_MFMA(P + 1, w, vc1, b1);
        SGV3D_SB();
    };
    for (int s = kb; s < ke; ++s) {
        const int nb = (s + 1 - kb) & 1;
        f32x4 *const a_wr = smem + nb * A_SLOTS + tid;
        const f32x4 *const Acur = smem + (nb ^ 1) * A_SLOTS + abase;
        const f32x4 *const Anew = smem + nb * A_SLOTS + abase;
        pair(std::integral_constant<int, 0>{}, va0, va1, vb0, vb1, Acur, Anew, a_wr);
        pair(std::integral_constant<int, 2>{}, vb0, vb1, va0, va1, Acur, Anew, a_wr);
        SGV3D_WINO_PUBLISH();
        if (s + 2 < ke) x_step += WK * sizeof(float);   // the patch fetched in pair 4 is step s+2's
        SGV3D_SB();
        pair(std::integral_constant<int, 4>{}, va0, va1, vb0, vb1, Acur, Anew, a_wr);
        pair(std::integral_constant<int, 6>{}, vb0, vb1, va0, va1, Acur, Anew, a_wr);
        w_cur = w_next;
        if (s + 2 < ke) w_next += W_STEP * 4;
    }
#undef SGV3D_WH_MFMA
#undef SGV3D_WH_T
#undef SGV3D_WH_RAW

    // ---- epilogue: partial output transform, exchange with the partner wave, common epilogue on output row dy = ph ----
    // address set-up of wino_epilogue, for 32 channels and one output row (dy = ph) per tile; the residual values are asked
    // for first, so that they arrive under the exchange
    const int col = tn2 * 32 + t;
    const bool partial = a.split_k > 1;
    const long long row0 = ((long long)img * a.out_h + oy0) * a.out_w + ox0;
    const char *ybase;
    unsigned pixstride, lane_off;
    float sc = 1.f, sh = 0.f, gt = 1.f, floor_ = -__builtin_inff();
    if (partial) {
        ybase = reinterpret_cast<const char *>(a.ws + ((size_t)blockIdx.y * a.M + row0) * a.N + tn2 * 32);
        pixstride = a.N * 4u;
        lane_off = t * 4u;
    } else {
        if (col < a.N) {
            if (a.scale) sc = a.scale[col];
            if (a.bias) sh = a.bias[col];
            if (a.gate) gt = a.gate[(size_t)img * a.cout + col];
        }
        if (a.relu) floor_ = 0.f;
        if (a.mode == SGV3D_CONV_NORMAL) {
            ybase = reinterpret_cast<const char *>(a.y + row0 * a.y_ld + a.y_coff + tn2 * 32);
            pixstride = a.y_ld * 4u;
            lane_off = t * 4u;
        } else if (a.mode == SGV3D_CONV_NCHW_OUT) {
            const long long hw = (long long)a.out_h * a.out_w;
            ybase = reinterpret_cast<const char *>(a.y + ((size_t)img * a.y_ld + a.y_coff + tn2 * 32) * hw + (long long)oy0 * a.out_w + ox0);
            pixstride = 4u;
            lane_off = (unsigned)(t * hw * 4);
        } else {  // GROUP_PLANES: [cout/g][M][g], g = a.ks
            const int grp0 = (tn2 * 32) / a.ks, grp = col / a.ks;
            ybase = reinterpret_cast<const char *>(a.y + ((size_t)grp0 * a.M + row0) * a.ks);
            pixstride = a.ks * 4u;
            lane_off = (unsigned)((((size_t)(grp - grp0) * a.M) * a.ks + (col - grp * a.ks)) * 4);
        }
    }
    const unsigned rowpitch = a.out_w * pixstride;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ybase, 0, (int)0xffffff00u, 0x00020000);
    const bool has_res = !partial && a.res != nullptr;
    const unsigned rpix = a.res_ld * 4u, rpitch = a.out_w * rpix;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_res ? a.res + row0 * a.res_ld + tn2 * 32 : a.zeros), 0, has_res ? (int)0xffffff00u : 0, 0x00020000);
    constexpr int RW2 = 2 * G::RW;                 // output rows per tile half
    const int oy_wave = oy0 + wm * RW2;
    // accumulator element e is tile tw = 8*(e>>2) + 4h + (e&3) of the tile half: row tw / TC, column tw % TC
    unsigned voff[TC / 2][2], roff[TC / 2][2];
#pragma unroll
    for (int c = 0; c < TC / 2; ++c)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = 8 * h + 2 * ((c >> 2) * 8 + (c & 3)) + dx;
            const bool ok = (col < a.N) & (ox0 + xx < a.out_w);
            voff[c][dx] = ok ? lane_off + (wm * RW2) * rowpitch + xx * pixstride : 0xffffffffu;
            roff[c][dx] = (ok && has_res) ? t * 4u + (wm * RW2) * rpitch + xx * rpix : 0xffffffffu;
        }
    float resv[16][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int yy = 2 * ((8 * (e >> 2)) / TC) + ph;
        const int c = ((e >> 2) % (TC / 8)) * 4 + (e & 3);
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)      // (no residual: zero-sized resource -> 0; rows below the image are not touched)
            resv[e][dx] = (oy_wave + yy < a.out_h)
                              ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, roff[c][dx], yy * rpitch, 0))
                              : 0.f;
    }

    __syncthreads();                                   // every wave is done with the patch buffers: they become the exchange area
    f32x2 *const xbuf = reinterpret_cast<f32x2 *>(smem);     // [wave][e][lane]
    float keep[16][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        // rows of M this wave holds: (acc[0..3][e]) = row 2 ph, (acc[4..7][e]) = row 2 ph + 1; column transform first
        const float ra0 = acc[0][e] + acc[1][e] + acc[2][e], ra1 = acc[1][e] - acc[2][e] - acc[3][e];
        const float rb0 = acc[4][e] + acc[5][e] + acc[6][e], rb1 = acc[5][e] - acc[6][e] - acc[7][e];
        // A^T = [1 1 1 0; 0 1 -1 -1]:  Y0 = M0 + M1 + M2,  Y1 = M1 - M2 - M3
        f32x2 send;
        if (ph == 0) {       // rows 0, 1: all of their share of Y0 stays, M1 goes to Y1
            keep[e][0] = ra0 + rb0; keep[e][1] = ra1 + rb1;
            send = f32x2{rb0, rb1};
        } else {             // rows 2, 3: -M2 - M3 of Y1 stays, M2 goes to Y0
            keep[e][0] = -ra0 - rb0; keep[e][1] = -ra1 - rb1;
            send = f32x2{ra0, ra1};
        }
        xbuf[(wave * 16 + e) * 64 + lane] = send;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const f32x2 got = xbuf[((wave ^ 1) * 16 + e) * 64 + lane];
        const int yy = 2 * ((8 * (e >> 2)) / TC) + ph;
        const int c = ((e >> 2) % (TC / 8)) * 4 + (e & 3);
        if (oy_wave + yy < a.out_h) {              // wave-uniform
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v = keep[e][dx] + got[dx];
                if (!partial) v = fmaxf(v * sc + sh + resv[e][dx], floor_) * gt;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, voff[c][dx], yy * rowpitch, 0);
            }
        }
    }
}

// Patch-resident variant for layers with few input channels and many output channels (the fused first
// layer of the 36 CenterHead branches: 64 -> 2304 at 256x256): a workgroup keeps the raw patch of ALL
// k-steps of its 16x16 block in LDS (18x18 x cin floats, 83 KB at cin = 64) and walks over a range of
// cout tiles.  Per cout tile there is no patch traffic and no barrier at all; the weight-fragment ring
// keeps prefetching across cout-tile boundaries (the packed weights of consecutive (tile, step) pairs are
// contiguous), so the only per-tile cost besides the MFMAs is the epilogue.  All workgroups walk the cout
// tiles in the same order, so the weight panel of the moment is shared through L2.
__global__ __launch_bounds__(256, 1) void conv_wino_resident_kernel(const ConvArgs a) {
    constexpr int TC = 8;
    using G = WinoGeom<TC>;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int ngroups = gridDim.x / a.tiles_m;
    const int tm = blockIdx.x % a.tiles_m, grp = blockIdx.x / a.tiles_m;
    const int tn_begin = (int)((long long)a.tiles_n * grp / ngroups);
    const int tn_end = (int)((long long)a.tiles_n * (grp + 1) / ngroups);
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * 16, ox0 = bx * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, t = lane & 31;
    const int nsteps = a.cin / WK;

    WinoStreams st;
    const unsigned x_bytes = (unsigned)((size_t)a.M * a.x_ld * sizeof(float));
    st.x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)x_bytes, 0x00020000);
    // ---- the whole patch, once ----------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 256 + tid;
        const int hh = s / G::PLANE, rem = s - hh * G::PLANE;
        const int row = rem / (2 * G::HALF), r2 = rem - row * (2 * G::HALF);
        const int par = r2 / G::HALF, ch = r2 - par * G::HALF;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + 2 * ch + par;
        const bool ok = (s < G::USED) & (ch < TC + 1) & (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
        const unsigned xo = ok ? (unsigned)((((size_t)(img * a.in_h + iy) * a.in_w + ix) * a.x_ld + a.x_coff + hh * 4) * sizeof(float))
                               : 0xfffffff0u - (unsigned)(a.cin * sizeof(float));
        for (int ks = 0; ks < nsteps; ++ks)
            smem[ks * A_SLOTS + s] = wino_buffer_load(st.x_rsrc, xo, (unsigned)(ks * WK * sizeof(float)));
    }
    const unsigned w_bytes = (unsigned)((size_t)a.tiles_n * nsteps * W_STEP * sizeof(float));
    st.w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)w_bytes, 0x00020000);
    st.w_cur = (unsigned)((size_t)tn_begin * nsteps * W_STEP * sizeof(float));
    st.w_next = st.w_cur + W_STEP * 4;       // reads past the last tile are out of range and return 0
    st.w_lane = (unsigned)(h * 64 + wn * 32 + t) * 16u;
    st.x0 = st.x1 = st.x2 = st.x_step = 0;
    st.a_wr = smem;

    const int abase = (h * G::ROWS + 2 * (wm * G::RW + t / TC)) * (2 * G::HALF) + t % TC;
    f32x4 tc[4][4], raw[4][4], wf[8], va0, va1, vb0, vb1;
#pragma unroll
    for (int p = 0; p < 6; ++p) wf[p] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_cur + p * (W_POS * 4));
    __syncthreads();
    {
        const f32x4 *const A = smem + abase;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = A[i * (2 * G::HALF) + (j & 1) * G::HALF + (j >> 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tc[0][j] = wino_bt<0>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[1][j] = wino_bt<1>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[2][j] = wino_bt<2>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            tc[3][j] = wino_bt<3>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
        }
        va0 = wino_bt<0>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
        va1 = wino_bt<1>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
    }

    for (int tn = tn_begin; tn < tn_end; ++tn) {
        f32x16 acc[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
        for (int s = 0; s < nsteps; ++s) {
            // the second half of the step reads the patch of the next step (step 0 again after the last)
            const f32x4 *const An = smem + (s + 1 < nsteps ? s + 1 : 0) * A_SLOTS + abase;
            wino_pair<0, TC, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<2, TC, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<4, TC, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<6, TC, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<8, TC, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<10, TC, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<12, TC, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<14, TC, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            st.w_cur = st.w_next;
            st.w_next += W_STEP * 4;
        }
        wino_epilogue<TC>(a, acc, tn, img, oy0, ox0, wm, wn, h, t);
    }
}

// ------------------------------------------------------------------------------------------------
// CenterHead branches fused: [3x3 64 -> 64 + BN + ReLU] and [3x3 64 -> c_k + bias] of all branches
// (mmdet3d SeparateHead, reached through layers/heads/bev_height_head.py:110) in one kernel, so the 64-
// channel hidden maps (604 MB per frame at cfg-2: the largest tensor of the model) never reach HBM.
//
// Patch-resident Winograd kernel as above; one cout tile = one branch (hidden_ch == 64).  Instead of
// storing the tile, the epilogue puts the block's 16x16 x 64 hidden values (BN + ReLU applied, zero
// outside the image) and the branch's final weights into LDS and runs the final 3x3 convolution from
// there: thread = output pixel, weights are broadcast LDS reads.  A block can only add the taps that fall
// inside its own 16x16 hidden pixels: it writes those partial sums (+ bias) for its own pixels to the
// output, and the partial sums it owes to the one-pixel ring AROUND the block (owned by the 8 neighbours)
// to a small workspace; head_ring_fixup_kernel then adds each border pixel's <= 3 ring contributions in a
// fixed order (N, S, W, E, NW, NE, SW, SE) -- deterministic, no atomics.
//
// LDS (160 KB): patch of all k-steps without pad slots (8 x 648 x 16 B = 81 KB; costs a 2-way conflict on a
// quarter of the patch reads), hidden tile [256 px + 1 zero px][68 floats] (17 16-B slots per pixel:
// conflict-free for thread = pixel), final weights of the branch [c][9 taps][64] (<= 9 KB).
constexpr int HID_LD = 68;
constexpr int HEAD_HALF = 9;                        // patch slots per (row, column parity): no padding
constexpr int HEAD_PLANE = 18 * 2 * HEAD_HALF;      // 324
constexpr int HEAD_PATCH_SLOTS = 2 * HEAD_PLANE;    // 648 per k-step
constexpr int HEAD_RING = 68;                       // ring pixels around a 16x16 block
constexpr int HEAD_MAX_OUT = 4;                     // output channels per branch

struct HeadArgs {
    const float *w2, *bias2;      // [total_out][3][3][64], [total_out]
    const int *out_begin;         // [tiles_n + 1] first output channel of each branch
    float *out;                   // NCHW [batch][total_out][H][W]
    float *ring;                  // [tiles_m][total_out][68]
    int total_out;
};

// Partial final-conv sums of local output pixel (oy, ox) in [-1, 16]^2 over the taps ky in [KY0, KY1],
// kx in [KX0, KX1] from the hidden pixels inside the block; a tap whose source pixel lies outside the block
// reads the zero pixel kept behind the tile (no divergence).  wts: the branch's weights in LDS.
template <int CB, int KY0, int KY1, int KX0, int KX1>
__device__ __forceinline__ void head_taps(const float *hid, const float *wts, int oy, int ox, float (&accf)[HEAD_MAX_OUT]) {
    // two partial sums per output (even / odd channel pairs): the products are natural register pairs of
    // the 16-byte LDS reads, so they map onto packed FMAs (2 per instruction) without operand shuffles
    f32x2 acc2[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc2[c] = f32x2{0.f, 0.f};
#pragma unroll 1
    for (int ky = KY0; ky <= KY1; ++ky)
#pragma unroll 1
        for (int kx = KX0; kx <= KX1; ++kx) {
            const int sy = oy + ky - 1, sx = ox + kx - 1;
            const bool in = ((unsigned)sy < 16u) & ((unsigned)sx < 16u);
            const f32x4 *hp = reinterpret_cast<const f32x4 *>(hid + (in ? sy * 16 + sx : 256) * HID_LD);
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(wts + (ky * 3 + kx) * 64);
#pragma unroll
            for (int qd = 0; qd < 16; ++qd) {
                const f32x4 hv = hp[qd];
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const f32x4 wv = wp[c * (9 * 16) + qd];
                    acc2[c] = hv.xy * wv.xy + acc2[c];
                    acc2[c] = hv.zw * wv.zw + acc2[c];
                }
            }
        }
#pragma unroll
    for (int c = 0; c < HEAD_MAX_OUT; ++c) accf[c] = 0.f;
#pragma unroll
    for (int c = 0; c < CB; ++c) accf[c] = acc2[c].x + acc2[c].y;
}

template <int KY0, int KY1, int KX0, int KX1>
__device__ __forceinline__ void head_taps_n(int cb, const float *hid, const float *wts, int oy, int ox,
                                            float (&accf)[HEAD_MAX_OUT]) {
    if (cb == 1) head_taps<1, KY0, KY1, KX0, KX1>(hid, wts, oy, ox, accf);
    else if (cb == 2) head_taps<2, KY0, KY1, KX0, KX1>(hid, wts, oy, ox, accf);
    else if (cb == 3) head_taps<3, KY0, KY1, KX0, KX1>(hid, wts, oy, ox, accf);
    else head_taps<4, KY0, KY1, KX0, KX1>(hid, wts, oy, ox, accf);
}

// Own pixels of the block, channel-split: wave w takes hidden channels [16w, 16w + 16) of all 256 pixels,
// a lane a vertical strip of 4 pixels (x = lane & 15, y = 4 (lane >> 4) .. +3).  The strip's 6 x 3 hidden
// neighbourhood is read once per 4 channels (consecutive lanes = consecutive pixels: conflict-free) and
// every broadcast weight read feeds 4 pixels -- a third of the LDS traffic of thread = pixel, which is what
// bounds this phase.  Partial sums of the 4 waves are added afterwards in fixed order (head kernel).
template <int CB>
__device__ __forceinline__ void head_own_ksplit(const float *hid, const float *wts, int wave, int lane,
                                                float (&part)[4][HEAD_MAX_OUT]) {
    f32x2 acc2[4][CB];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc2[r][c] = f32x2{0.f, 0.f};
    const int x = lane & 15, y0 = 4 * (lane >> 4);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int qd = wave * 4 + q;
        f32x4 hv[6][3];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sy = y0 - 1 + i, sx = x - 1 + j;
                const bool in = ((unsigned)sy < 16u) & ((unsigned)sx < 16u);
                hv[i][j] = reinterpret_cast<const f32x4 *>(hid + (in ? sy * 16 + sx : 256) * HID_LD)[qd];
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const f32x4 wv = reinterpret_cast<const f32x4 *>(wts + (c * 9 + ky * 3 + kx) * 64)[qd];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc2[r][c] = hv[r + ky][kx].xy * wv.xy + acc2[r][c];
                        acc2[r][c] = hv[r + ky][kx].zw * wv.zw + acc2[r][c];
                    }
                }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < HEAD_MAX_OUT; ++c) part[r][c] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) part[r][c] = acc2[r][c].x + acc2[r][c].y;
}

__global__ __launch_bounds__(256, 1) void conv_wino_head_kernel(const ConvArgs a, const HeadArgs ha) {
    constexpr int TC = 8;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int nsteps = a.cin / WK;
    float *const hid = reinterpret_cast<float *>(smem + nsteps * HEAD_PATCH_SLOTS);
    float *const wts = hid + 257 * HID_LD;
    const int tm = blockIdx.x;
    const int bpi = a.wb_y * a.wb_x;
    const int img = tm / bpi;
    const int rb = tm - img * bpi;
    const int by = rb / a.wb_x, bx = rb - by * a.wb_x;
    const int oy0 = by * 16, ox0 = bx * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, t = lane & 31;

    WinoStreams st;
    const unsigned x_bytes = (unsigned)((size_t)a.M * a.x_ld * sizeof(float));
    st.x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)x_bytes, 0x00020000);
    // ---- the whole patch, once: slot -> (channel half, row, column parity, column / 2) ------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 256 + tid;
        const int hh = s / HEAD_PLANE, rem = s - hh * HEAD_PLANE;
        const int row = rem / (2 * HEAD_HALF), r2 = rem - row * (2 * HEAD_HALF);
        const int par = r2 / HEAD_HALF, ch = r2 - par * HEAD_HALF;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + 2 * ch + par;
        const bool ok = (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
        const unsigned xo = ok ? (unsigned)((((size_t)(img * a.in_h + iy) * a.in_w + ix) * a.x_ld + a.x_coff + hh * 4) * sizeof(float))
                               : 0xfffffff0u - (unsigned)(a.cin * sizeof(float));
        if (s < HEAD_PATCH_SLOTS)
            for (int ks = 0; ks < nsteps; ++ks)
                smem[ks * HEAD_PATCH_SLOTS + s] = wino_buffer_load(st.x_rsrc, xo, (unsigned)(ks * WK * sizeof(float)));
    }
    const unsigned w_bytes = (unsigned)((size_t)a.tiles_n * nsteps * W_STEP * sizeof(float));
    st.w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)w_bytes, 0x00020000);
    st.w_cur = st.w_next = 0;
    st.w_lane = (unsigned)(h * 64 + wn * 32 + t) * 16u;
    st.x0 = st.x1 = st.x2 = st.x_step = 0;
    st.a_wr = smem;
    if (tid < HID_LD) hid[256 * HID_LD + tid] = 0.f;   // the zero pixel out-of-block taps read

    const int abase = (h * 18 + 2 * (wm * 4 + (t >> 3))) * (2 * HEAD_HALF) + (t & 7);
    __syncthreads();

    const size_t plane = (size_t)a.out_h * a.out_w;
    for (int tn = 0; tn < a.tiles_n; ++tn) {
        // Nothing of the main loop's register state lives across the epilogue: fragment ring, transformed
        // patch of step 0 and accumulators are set up per branch.
        f32x4 tc[4][4], raw[4][4], wf[8], va0, va1, vb0, vb1;
        st.w_cur = (unsigned)((size_t)tn * nsteps * W_STEP * sizeof(float));
        st.w_next = st.w_cur + W_STEP * 4;       // reads past the last tile are out of range and return 0
#pragma unroll
        for (int p = 0; p < 6; ++p) wf[p] = wino_buffer_load(st.w_rsrc, st.w_lane, st.w_cur + p * (W_POS * 4));
        {
            const f32x4 *const A = smem + abase;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[i][j] = A[i * (2 * HEAD_HALF) + (j & 1) * HEAD_HALF + (j >> 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tc[0][j] = wino_bt<0>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
                tc[1][j] = wino_bt<1>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
                tc[2][j] = wino_bt<2>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
                tc[3][j] = wino_bt<3>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
            }
            va0 = wino_bt<0>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
            va1 = wino_bt<1>(tc[0][0], tc[0][1], tc[0][2], tc[0][3]);
        }
        f32x16 acc[16];                  // (not cleared: the first k-step's MFMAs start from the constant 0)
        {
            const f32x4 *const An = smem + (1 < nsteps ? 1 : 0) * HEAD_PATCH_SLOTS + abase;
            wino_pair<0, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<2, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<4, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<6, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<8, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<10, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<12, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<14, TC, true, 3, HEAD_HALF, true>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            st.w_cur = st.w_next;
            st.w_next += W_STEP * 4;
        }
        for (int s = 1; s < nsteps; ++s) {
            const f32x4 *const An = smem + (s + 1 < nsteps ? s + 1 : 0) * HEAD_PATCH_SLOTS + abase;
            wino_pair<0, TC, true, 3, HEAD_HALF>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<2, TC, true, 3, HEAD_HALF>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<4, TC, true, 3, HEAD_HALF>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<6, TC, true, 3, HEAD_HALF>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<8, TC, true, 3, HEAD_HALF>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<10, TC, true, 3, HEAD_HALF>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            wino_pair<12, TC, true, 3, HEAD_HALF>(acc, tc, raw, va0, va1, vb0, vb1, wf, st, An);
            wino_pair<14, TC, true, 3, HEAD_HALF>(acc, tc, raw, vb0, vb1, va0, va1, wf, st, An);
            st.w_cur = st.w_next;
            st.w_next += W_STEP * 4;
        }
        // ---- hidden tile + the branch's final weights -> LDS ------------------------------------------------
        SGV3D_WINO_PUBLISH();          // every wave is done reading the previous branch's hidden tile / weights
        // Opaque copies of the thread / block coordinates: what the epilogue derives from them (bounds masks,
        // LDS addresses, output pointers) is recomputed per branch instead of being hoisted out of the branch
        // loop and kept -- or spilled -- across the main loop, which has no registers to spare.
        int tid_ = tid, oy0_ = oy0, ox0_ = ox0;
        asm volatile("" : "+v"(tid_), "+s"(oy0_), "+s"(ox0_));
        const int lane_ = tid_ & 63, wave_ = __builtin_amdgcn_readfirstlane(tid_ >> 6);
        const int wm_ = wave_ >> 1, wn_ = wave_ & 1, h_ = lane_ >> 5, t_ = lane_ & 31;
        const int o0 = __builtin_amdgcn_readfirstlane(ha.out_begin[tn]);
        const int cb = __builtin_amdgcn_readfirstlane(min(ha.out_begin[tn + 1] - o0, HEAD_MAX_OUT));
        if (tid_ < cb * 144)            // cb x 9 x 64 floats, contiguous in w2
            reinterpret_cast<f32x4 *>(wts)[tid_] = reinterpret_cast<const f32x4 *>(ha.w2 + (size_t)o0 * 576)[tid_];
        if (tid_ + 256 < cb * 144)
            reinterpret_cast<f32x4 *>(wts)[tid_ + 256] = reinterpret_cast<const f32x4 *>(ha.w2 + (size_t)o0 * 576)[tid_ + 256];
        if (tid_ + 512 < cb * 144)
            reinterpret_cast<f32x4 *>(wts)[tid_ + 512] = reinterpret_cast<const f32x4 *>(ha.w2 + (size_t)o0 * 576)[tid_ + 512];
        {
            const int col = tn * 64 + wn_ * 32 + t_;
            const float sc = a.scale ? a.scale[col] : 1.f, sh = a.bias ? a.bias[col] : 0.f;
            float *const hl = hid + (wn_ * 32 + t_);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float r0[4], r1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m0 = acc[j][e], m1 = acc[4 + j][e], m2 = acc[8 + j][e], m3 = acc[12 + j][e];
                    r0[j] = m0 + m1 + m2;
                    r1[j] = m1 - m2 - m3;
                }
                const float yv[2][2] = {{r0[0] + r0[1] + r0[2], r0[1] - r0[2] - r0[3]},
                                        {r1[0] + r1[1] + r1[2], r1[1] - r1[2] - r1[3]}};
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int ly = 2 * (wm_ * 4 + (e >> 2)) + dy, lx = 2 * (4 * h_ + (e & 3)) + dx;
                        const bool in = (oy0_ + ly < a.out_h) & (ox0_ + lx < a.out_w);
                        hl[(ly * 16 + lx) * HID_LD] = in ? fmaxf(yv[dy][dx] * sc + sh, 0.f) : 0.f;
                    }
            }
        }
        __syncthreads();               // (also waits for the weight loads above)
        // ---- final 3x3 conv of this branch from LDS -------------------------------------------------------
        float accf[HEAD_MAX_OUT];
        {   // the ring owed to the neighbours: one side per wave, only the three taps that can reach the block
            int ridx;
            if (wave_ == 0) { ridx = lane_; head_taps_n<2, 2, 0, 2>(cb, hid, wts, -1, lane_ - 1, accf); }
            else if (wave_ == 1) { ridx = 18 + lane_; head_taps_n<0, 0, 0, 2>(cb, hid, wts, 16, lane_ - 1, accf); }
            else if (wave_ == 2) { ridx = 36 + lane_; head_taps_n<0, 2, 2, 2>(cb, hid, wts, lane_, -1, accf); }
            else { ridx = 52 + lane_; head_taps_n<0, 2, 0, 0>(cb, hid, wts, lane_, 16, accf); }
            if (lane_ < (wave_ < 2 ? 18 : 16)) {
                float *rp = ha.ring + ((size_t)tm * ha.total_out + o0) * HEAD_RING + ridx;
#pragma unroll
                for (int c = 0; c < HEAD_MAX_OUT; ++c)
                    if (c < cb) rp[c * HEAD_RING] = accf[c];
            }
        }
        {   // own pixels, channel-split over the waves; partial sums meet in LDS (over the hidden tile)
            float part[4][HEAD_MAX_OUT];
            if (cb == 1) head_own_ksplit<1>(hid, wts, wave_, lane_, part);
            else if (cb == 2) head_own_ksplit<2>(hid, wts, wave_, lane_, part);
            else if (cb == 3) head_own_ksplit<3>(hid, wts, wave_, lane_, part);
            else head_own_ksplit<4>(hid, wts, wave_, lane_, part);
            SGV3D_WINO_PUBLISH();      // every wave is done reading the hidden tile
            f32x4 *const pl = reinterpret_cast<f32x4 *>(hid);           // [wave][pixel] x 4 outputs
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pl[wave_ * 256 + (4 * (lane_ >> 4) + r) * 16 + (lane_ & 15)] = f32x4{part[r][0], part[r][1], part[r][2], part[r][3]};
            SGV3D_WINO_PUBLISH();
            const int py = tid_ >> 4, px = tid_ & 15;
            const f32x4 s4 = ((pl[tid_] + pl[256 + tid_]) + pl[512 + tid_]) + pl[768 + tid_];
            if (oy0_ + py < a.out_h && ox0_ + px < a.out_w) {
                float *op = ha.out + ((size_t)img * ha.total_out + o0) * plane + (size_t)(oy0_ + py) * a.out_w + ox0_ + px;
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int c = 0; c < HEAD_MAX_OUT; ++c)
                    if (c < cb) op[c * plane] = sv[c] + ha.bias2[o0 + c];
            }
        }
    }
}

// out[b][oc][y][x] += ring contributions of the up-to-3 neighbouring blocks, fixed order N S W E NW NE SW SE
__global__ void head_ring_fixup_kernel(int batch, int H, int W, int nby, int nbx, int total_out,
                                       const float *__restrict__ ring, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)batch * nby * nbx * total_out * 60;
    if (i >= total) return;
    const int bi = (int)(i % 60);
    long long r = i / 60;
    const int oc = (int)(r % total_out);
    r /= total_out;
    const int bx = (int)(r % nbx);
    r /= nbx;
    const int by = (int)(r % nby), img = (int)(r / nby);
    int py, px;   // the 60 border pixels of a 16x16 block
    if (bi < 16) { py = 0; px = bi; }
    else if (bi < 32) { py = 15; px = bi - 16; }
    else if (bi < 46) { py = bi - 32 + 1; px = 0; }
    else { py = bi - 46 + 1; px = 15; }
    const int y = by * 16 + py, x = bx * 16 + px;
    if (y >= H || x >= W) return;
    auto R = [&](int nby_, int nbx_, int ridx) -> float {
        if (nby_ < 0 || nby_ >= nby || nbx_ < 0 || nbx_ >= nbx) return 0.f;
        const size_t blk = ((size_t)img * nby + nby_) * nbx + nbx_;
        return ring[(blk * total_out + oc) * HEAD_RING + ridx];
    };
    float v = out[((size_t)img * total_out + oc) * H * W + (size_t)y * W + x];
    if (py == 0) v += R(by - 1, bx, 18 + px + 1);
    if (py == 15) v += R(by + 1, bx, px + 1);
    if (px == 0) v += R(by, bx - 1, 52 + py);
    if (px == 15) v += R(by, bx + 1, 36 + py);
    if (py == 0 && px == 0) v += R(by - 1, bx - 1, 35);
    if (py == 0 && px == 15) v += R(by - 1, bx + 1, 18);
    if (py == 15 && px == 0) v += R(by + 1, bx - 1, 17);
    if (py == 15 && px == 15) v += R(by + 1, bx + 1, 0);
    out[((size_t)img * total_out + oc) * H * W + (size_t)y * W + x] = v;
}

// U = G g G^T per (cout, cin), written in the order the kernel streams it:
// [cout tile of 64][k-step of 8 channels][pos 16][channel half 2][n 64][4 channels]
__global__ void wino_pack_weight_kernel(const float *__restrict__ src, int cout, int cin, int cin_pad,
                                        float *__restrict__ dst, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 3);
    const int n = (int)((i >> 2) & 63);
    const int hh = (int)((i >> 8) & 1);
    const int pos = (int)((i >> 9) & 15);
    const long long st = i >> 13;
    const int nsteps = cin_pad / WK;
    const int step = (int)(st % nsteps), tn = (int)(st / nsteps);
    const int co = tn * 64 + n, ci = step * WK + hh * 4 + j;
    float v = 0.f;
    if (co < cout && ci < cin) {
        const float *g = src + ((size_t)co * cin + ci) * 9;
        const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
        const int pi = pos >> 2, pj = pos & 3;
        double u = 0.0;
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x) u += G[pi][y] * (double)g[y * 3 + x] * G[pj][x];
        v = (float)u;
    }
    dst[i] = v;
}

}  // namespace

extern "C" size_t sgv3d_conv_winograd_weight_floats(int cout, int cin_pad) {
    if (cout <= 0 || cin_pad <= 0 || cin_pad % WK) return 0;
    return (size_t)((cout + 63) / 64) * (cin_pad / WK) * W_STEP;
}

extern "C" int sgv3d_conv_winograd_pack_weight(const float *w_src, int cout, int cin, int cin_pad, float *w_packed,
                                               void *stream) {
    SGV3D_REQUIRE(w_src && w_packed, "conv_winograd_pack_weight: null pointer");
    SGV3D_REQUIRE(cout > 0 && cin > 0 && cin_pad >= cin && cin_pad % WK == 0,
                  "conv_winograd_pack_weight: cin_pad=%d must cover cin=%d and be a multiple of 8", cin_pad, cin);
    const long long total = (long long)sgv3d_conv_winograd_weight_floats(cout, cin_pad);
    hipLaunchKernelGGL(wino_pack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, cout,
                       cin, cin_pad, w_packed, total);
    return check_launch("wino_pack_weight_kernel");
}

extern "C" int sgv3d_conv2d_winograd_forward(const sgv3d_conv_desc *d, const float *x, const float *w_wino,
                                             const float *scale, const float *bias, const float *residual,
                                             const float *gate, float *y, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    SGV3D_REQUIRE(d && x && w_wino && y, "conv2d_winograd_forward: null pointer");
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1,
                  "conv2d_winograd_forward: only 3x3 / stride 1 / dilation 1 / pad 1 (got k%dx%d s%d d%d p%d)", d->kh,
                  d->kw, d->stride, d->dil, d->pad);
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0, "conv2d_winograd_forward: non-positive dimension");
    SGV3D_REQUIRE(d->out_h == d->in_h && d->out_w == d->in_w, "conv2d_winograd_forward: output must equal input size");
    SGV3D_REQUIRE(d->cin % WK == 0 && (d->x_ld & 3) == 0 && (d->x_coff & 3) == 0,
                  "conv2d_winograd_forward: cin must be a multiple of 8, x_ld/x_coff of 4 (got %d/%d/%d)", d->cin, d->x_ld, d->x_coff);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_wino) & 15) == 0,
                  "conv2d_winograd_forward: x and w_wino must be 16-B aligned");
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin, "conv2d_winograd_forward: x_ld too small");
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL || d->mode == SGV3D_CONV_NCHW_OUT || d->mode == SGV3D_CONV_GROUP_PLANES,
                  "conv2d_winograd_forward: bad mode %d", d->mode);
    SGV3D_REQUIRE(d->mode != SGV3D_CONV_GROUP_PLANES || (d->deconv_ks > 0 && d->cout % d->deconv_ks == 0 && !residual),
                  "conv2d_winograd_forward: GROUP_PLANES needs deconv_ks = group width dividing cout, no residual");
    SGV3D_REQUIRE(d->mode != SGV3D_CONV_NCHW_OUT || residual == nullptr, "conv2d_winograd_forward: NCHW_OUT has no residual");
    SGV3D_REQUIRE(d->mode != SGV3D_CONV_NORMAL || d->y_ld >= d->y_coff + d->cout, "conv2d_winograd_forward: y_ld too small");
    SGV3D_REQUIRE(residual == nullptr || d->res_ld >= d->cout, "conv2d_winograd_forward: res_ld too small");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    SGV3D_REQUIRE(M < 0x7fffffffLL, "conv2d_winograd_forward: too many output pixels");
    SGV3D_REQUIRE((long long)d->batch * d->in_h * d->in_w * d->x_ld * 4 < 0xf0000000LL,
                  "conv2d_winograd_forward: input larger than 3.75 GiB (32-bit buffer offsets)");
    SGV3D_REQUIRE((long long)sgv3d_conv_winograd_weight_floats(d->cout, d->cin) * 4 < 0xf0000000LL,
                  "conv2d_winograd_forward: packed weights larger than 3.75 GiB");
    ConvArgs a;
    a.x = x; a.w = w_wino; a.scale = scale; a.bias = bias; a.res = residual; a.gate = gate; a.y = y;
    a.zeros = conv_zero_block();
    if (!a.zeros) return fail(SGV3D_ELAUNCH, "conv2d_winograd_forward: cannot resolve the zero block");
    a.M = (int)M; a.N = d->cout; a.K = 9 * d->cin; a.k_pad = a.K;
    a.in_h = d->in_h; a.in_w = d->in_w; a.cin = d->cin; a.out_h = d->out_h; a.out_w = d->out_w; a.cout = d->cout;
    a.m_h = d->out_h; a.m_w = d->out_w;
    a.kh = 3; a.kw = 3; a.stride = 1; a.pad = 1; a.dil = 1;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld;
    a.relu = d->relu; a.mode = d->mode; a.ks = d->deconv_ks;
    a.korder = 0;
    // tile-block shape: 8x8 tiles (16x16 pixels) or 4x16 tiles (8x32 pixels), whichever covers the image
    // with fewer blocks (54x96: 21 instead of 24); the patch-resident variant is built for 8x8 only
    const int blocks8 = cdiv(d->out_h, 16) * cdiv(d->out_w, 16), blocks16 = cdiv(d->out_h, 8) * cdiv(d->out_w, 32);
    const bool wide = blocks16 < blocks8 && d->tile != SGV3D_WINOGRAD_RESIDENT;
    a.wb_y = wide ? cdiv(d->out_h, 8) : cdiv(d->out_h, 16);
    a.wb_x = wide ? cdiv(d->out_w, 32) : cdiv(d->out_w, 16);
    a.tiles_m = d->batch * a.wb_y * a.wb_x;
    a.tiles_n = cdiv(d->cout, 64);
    a.split_k = d->split_k > 1 ? d->split_k : 1;
    a.ws = static_cast<float *>(workspace);
    SGV3D_REQUIRE(a.split_k <= d->cin / WK && a.split_k <= 64, "conv2d_winograd_forward: split_k=%d too large for %d k-steps",
                  a.split_k, d->cin / WK);
    if (a.split_k > 1) {
        const size_t need = sizeof(float) * (size_t)a.split_k * a.M * a.N;
        if (!workspace || workspace_bytes < need)
            return fail(SGV3D_ENOSPACE, "conv2d_winograd_forward: split-K workspace has %zu bytes, needs %zu", workspace_bytes, need);
    }
    hipStream_t st = as_stream(stream);
    if (d->tile == SGV3D_WINOGRAD_RESIDENT) {
        // whole patch in LDS: cin/8 steps x 12 KB
        const int lds = (d->cin / WK) * A_SLOTS * 16;
        SGV3D_REQUIRE(lds <= 160 * 1024 - 4096, "conv2d_winograd_forward: patch-resident variant needs cin <= 96 (got %d)", d->cin);
        SGV3D_REQUIRE(a.split_k == 1, "conv2d_winograd_forward: patch-resident variant has no split-K");
        static PerDeviceSize lds_set;
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wino_resident_kernel), (size_t)lds, lds_set))
            return fail(SGV3D_ELAUNCH, "conv2d_winograd_forward: cannot raise the dynamic LDS limit to %d", lds);
        // cout tiles are dealt to `groups` workgroups per block so that the grid is about one wave of CUs
        int groups = 256 / a.tiles_m;
        if (groups < 1) groups = 1;
        if (groups > a.tiles_n) groups = a.tiles_n;
        hipLaunchKernelGGL(conv_wino_resident_kernel, dim3(a.tiles_m * groups), dim3(256), lds, st, a);
        return check_launch("conv_wino_resident_kernel");
    }
    if (d->tile == SGV3D_WINOGRAD_HALF) {           // 64 tiles x 32 channels per workgroup, two workgroups per CU
        a.tiles_n = cdiv(d->cout, 32);
        if (wide) hipLaunchKernelGGL(conv_wino_half_kernel<16>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoHalfLds, st, a);
        else hipLaunchKernelGGL(conv_wino_half_kernel<8>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoHalfLds, st, a);
        if (a.split_k > 1) return launch_splitk_reduce(a, st);
        return check_launch("conv_wino_half_kernel");
    }
    if (wide) hipLaunchKernelGGL(conv_wino_kernel<16>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoLds, st, a);
    else hipLaunchKernelGGL(conv_wino_kernel<8>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoLds, st, a);
    if (a.split_k > 1) return launch_splitk_reduce(a, st);
    return check_launch("conv_wino_kernel");
}

namespace sgv3d {
// Second launch of both fused CenterHead kernels (conv_wino_head_kernel here, head_wino4_kernel in head_wino4.hip): adds the
// ring partial sums [blocks][total_out][68] of the neighbouring 16x16 blocks to the border pixels of `out`.
int launch_head_ring_fixup(int batch, int h, int w, int total_out, const float *ring, float *out, hipStream_t st) {
    const int nby = cdiv(h, 16), nbx = cdiv(w, 16);
    const long long total = (long long)batch * nby * nbx * total_out * 60;
    hipLaunchKernelGGL(head_ring_fixup_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, batch, h, w, nby, nbx, total_out, ring, out);
    return check_launch("head_ring_fixup_kernel");
}
}  // namespace sgv3d

extern "C" size_t sgv3d_centerhead_branches_workspace_bytes(int batch, int h, int w, int total_out) {
    if (batch <= 0 || h <= 0 || w <= 0 || total_out <= 0) return 0;
    return sizeof(float) * (size_t)batch * cdiv(h, 16) * cdiv(w, 16) * total_out * HEAD_RING;
}

extern "C" int sgv3d_centerhead_branches_forward(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                                 int num_branches, const float *w1_wino, const float *scale1,
                                                 const float *bias1, int total_out, const float *w2,
                                                 const float *bias2, const int32_t *out_begin, float *out,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(x && w1_wino && w2 && bias2 && out_begin && out, "centerhead_branches_forward: null pointer");
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && num_branches > 0 && total_out > 0, "centerhead_branches_forward: non-positive dimension");
    SGV3D_REQUIRE(cin % WK == 0 && (x_ld & 3) == 0 && (x_coff & 3) == 0 && x_ld >= x_coff + cin,
                  "centerhead_branches_forward: cin must be a multiple of 8, x_ld/x_coff of 4");
    const int lds = (cin / WK) * HEAD_PATCH_SLOTS * 16 + 257 * HID_LD * 4 + HEAD_MAX_OUT * 576 * 4;
    SGV3D_REQUIRE(lds <= 160 * 1024, "centerhead_branches_forward: cin=%d too large for the patch-resident kernel (<= 64)", cin);
    SGV3D_REQUIRE((long long)batch * h * w * x_ld * 4 < 0xf0000000LL, "centerhead_branches_forward: input larger than 3.75 GiB");
    const size_t need = sgv3d_centerhead_branches_workspace_bytes(batch, h, w, total_out);
    if (!workspace || workspace_bytes < need)
        return fail(SGV3D_ENOSPACE, "centerhead_branches_forward: workspace has %zu bytes, needs %zu", workspace_bytes, need);
    ConvArgs a;
    a.x = x; a.w = w1_wino; a.scale = scale1; a.bias = bias1; a.res = nullptr; a.gate = nullptr; a.y = nullptr;
    a.zeros = nullptr; a.ws = nullptr;
    a.M = batch * h * w; a.N = num_branches * 64; a.K = 9 * cin; a.k_pad = a.K;
    a.in_h = h; a.in_w = w; a.cin = cin; a.out_h = h; a.out_w = w; a.cout = a.N;
    a.m_h = h; a.m_w = w; a.kh = 3; a.kw = 3; a.stride = 1; a.pad = 1; a.dil = 1;
    a.x_ld = x_ld; a.x_coff = x_coff; a.y_ld = 0; a.y_coff = 0; a.res_ld = 0; a.relu = 1; a.mode = 0; a.ks = 0;
    a.korder = 0; a.split_k = 1;
    a.wb_y = cdiv(h, 16); a.wb_x = cdiv(w, 16);
    a.tiles_m = batch * a.wb_y * a.wb_x;
    a.tiles_n = num_branches;
    HeadArgs ha;
    ha.w2 = w2; ha.bias2 = bias2; ha.out_begin = out_begin; ha.out = out; ha.ring = static_cast<float *>(workspace);
    ha.total_out = total_out;
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_wino_head_kernel), (size_t)lds, lds_set))
        return fail(SGV3D_ELAUNCH, "centerhead_branches_forward: cannot raise the dynamic LDS limit to %d", lds);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(conv_wino_head_kernel, dim3(a.tiles_m), dim3(256), lds, st, a, ha);
    const int rc = launch_head_ring_fixup(batch, h, w, total_out, ha.ring, out, st);
    if (rc != SGV3D_OK) return rc;
    return check_launch("conv_wino_head_kernel");
}
