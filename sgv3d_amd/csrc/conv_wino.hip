// Winograd F(2x2, 3x3) convolution for gfx950, fp32 in / fp32 accumulate, NHWC activations:
// the 3x3 / stride-1 / pad-1 convolutions of the path (ResNet 3x3s, HeightNet BasicBlocks, BEV trunk,
// CenterHead shared + fused branch layer) at 2.25x fewer MFMA flops than the implicit GEMM.  cuDNN
// picks the same algorithm family for these layers in the reference (layers/backbones/lss_fpn.py:
// 186-198,207-250; layers/heads/bev_height_head.py:75-110), so the rounding class is the reference's.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input tile d
//
// One kernel, nothing spilled to HBM in the Winograd domain:
//   * workgroup = 4 waves (2 x 2) = 64 tiles (an 8x8 block = 16x16 output pixels) x 64 output channels
//     x all 16 Winograd positions; each wave owns 32 tiles x 32 channels x 16 positions = 16 MFMA
//     accumulator tiles (256 accumulator registers: one wave per SIMD, the whole 512-register file).
//   * k-step = 8 input channels.  Per step the 18x18 raw input patch (11.5 KB) and the pre-transformed
//     weights of the step ([16 pos][2][64 n][4], 32 KB, one linear stream thanks to the packing) are
//     copied global -> LDS by global_load_lds_dwordx4 (no staging registers), double-buffered.
//   * the input transform B^T d B runs in registers on the fly (each lane reads the 16 raw float4 of its
//     tile once per step: 64 add/sub per component), feeding 16 x 4 MFMAs per step and wave.
//   * LDS images are conflict-free: the patch is stored [k/4][row][column parity][column/2 (pad 10)] so
//     the 16 lanes of a ds_read_b128 group (8 tiles x 2 tile rows) hit 16 distinct 16-B slots.
//   * the output transform A^T M A is lane-local (the 16 positions of one (tile, channel) element sit in
//     the same lane / register index of the 16 accumulator tiles), followed by the common epilogue.
//   * split-K over input-channel steps (grid.y) for layers with too few tiles to fill 256 CUs; partial
//     OUTPUT tiles (the transform is linear) go to the workspace and the igemm reduce kernel finishes.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
template <int P, int TC, bool RESIDENT = false>
__device__ __forceinline__ void wino_pair(f32x16 (&acc)[16], f32x4 (&tc)[4][4], f32x4 (&raw)[4][4], f32x4 &vc0,
                                          f32x4 &vc1, f32x4 &vn0, f32x4 &vn1, f32x4 (&wf)[8], WinoStreams &st,
                                          const f32x4 *An) {
    static_assert((P & 1) == 0, "pairs start at even positions");
    constexpr int R0 = P & 7, R1 = (P + 1) & 7;          // ring slots of this pair
    constexpr int L0 = (P + 6) & 7, L1 = (P + 7) & 7;    // ring slots (= those of pair P-2) refilled now
    const f32x4 b0 = wf[R0], b1 = wf[R1];
    SGV3D_WINO_MFMA(P, x, vc0, b0);
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
                raw[i][j] = An[i * (2 * WinoGeom<TC>::HALF) + (j & 1) * WinoGeom<TC>::HALF + (j >> 1)];
    } else if constexpr (!RESIDENT && P == 10) {
        st.stage0 = wino_buffer_load(st.x_rsrc, st.x0, st.x_step);
    } else if constexpr (!RESIDENT && P == 12) {
        st.stage1 = wino_buffer_load(st.x_rsrc, st.x1, st.x_step);
    } else if constexpr (!RESIDENT && P == 14) {
        st.stage2 = wino_buffer_load(st.x_rsrc, st.x2, st.x_step);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, x, vc1, b1);
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
        static int lds_set = 0;
        if (lds > lds_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino_resident_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return fail(SGV3D_ELAUNCH, "conv2d_winograd_forward: cannot raise the dynamic LDS limit to %d", lds);
            lds_set = lds;
        }
        // cout tiles are dealt to `groups` workgroups per block so that the grid is about one wave of CUs
        int groups = 256 / a.tiles_m;
        if (groups < 1) groups = 1;
        if (groups > a.tiles_n) groups = a.tiles_n;
        hipLaunchKernelGGL(conv_wino_resident_kernel, dim3(a.tiles_m * groups), dim3(256), lds, st, a);
        return check_launch("conv_wino_resident_kernel");
    }
    if (wide) hipLaunchKernelGGL(conv_wino_kernel<16>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoLds, st, a);
    else hipLaunchKernelGGL(conv_wino_kernel<8>, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoLds, st, a);
    if (a.split_k > 1) return launch_splitk_reduce(a, st);
    return check_launch("conv_wino_kernel");
}
