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
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

constexpr int WK = 8;                           // input channels per k-step
constexpr int A_ROWS = 18, A_HALF = 10;         // patch rows; 16-B slots per (row, column parity), 9 used
constexpr int A_PLANE = A_ROWS * 2 * A_HALF;    // 360 slots per 4-channel plane
constexpr int A_USED = 2 * A_PLANE;             // 720
constexpr int A_SLOTS = 768;                    // 12 wave-instructions of 64 slots
constexpr int W_SLOTS = 16 * 2 * 64;            // 2048 slots = 32 KB per step
constexpr int BUF_SLOTS = A_SLOTS + W_SLOTS;    // 45 056 B per buffer
constexpr int kWinoLds = 3 * BUF_SLOTS * 16;           // 135 168 B: three buffers

// Asynchronous 16-byte-per-lane global -> LDS copy (lane i lands at dst_wave_base + 16*i).  Issued as
// inline asm: the compiler would otherwise drain it (vmcnt(0)) before the next ds_read.  The copies of
// step s+2 are issued right after the barrier in the middle of step s and retired by the
// "s_waitcnt vmcnt(0); s_barrier" in the middle of step s+1 (SGV3D_WINO_PUBLISH), one full step later.
__device__ __forceinline__ void glds16(const float *src, float4 *dst_wave_base) {
    unsigned keep;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void *)dst_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
#define SGV3D_WINO_PUBLISH() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
#define SGV3D_SB() __builtin_amdgcn_sched_barrier(0)

// The 11 copies (3 patch pieces + 8 weight pieces of 1 KiB) a wave issues per k-step.  An LDS-DMA piece
// costs the issuing wave ~60-180 cycles of vector issue, so the pieces are not issued in a burst after the
// barrier but one per MFMA gap in the second half of the step (wino_pair).
struct WinoLoader {
    const float *asrc[3];
    int ainc[3];
    const float *wsrc;
    float4 *dst;      // destination buffer (LDS) of the step being fetched
    int wave;
    bool on;          // false in the last two steps: nothing left to fetch
    template <int K>
    __device__ __forceinline__ void piece() {
        if (!on) return;
        if constexpr (K < 3) {
            glds16(asrc[K], dst + (wave * 3 + K) * 64);
            asrc[K] += ainc[K];
        } else {
            glds16(wsrc + (K - 3) * 256, dst + A_SLOTS + (wave * 8 + (K - 3)) * 64);
            if constexpr (K == 10) wsrc += W_SLOTS * 4;
        }
    }
    __device__ __forceinline__ void all() {
        piece<0>(); piece<1>(); piece<2>(); piece<3>(); piece<4>(); piece<5>();
        piece<6>(); piece<7>(); piece<8>(); piece<9>(); piece<10>();
    }
};

// B^T rows of F(2x2,3x3): (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
template <int R>
__device__ __forceinline__ float4 wino_bt(const float4 &x0, const float4 &x1, const float4 &x2, const float4 &x3) {
    if constexpr (R == 0) return x0 - x2;
    else if constexpr (R == 1) return x1 + x2;
    else if constexpr (R == 2) return x2 - x1;
    else return x1 - x3;
}

// Two Winograd positions P, P+1 of the current step: 2 x 4 MFMAs (k = 4h + 0..3), interleaved so that
// every MFMA is followed by one on the other accumulator, with the rest of the pipeline in their shadows
// (the wave is alone on its SIMD, so whatever is issued between two MFMAs is free while the issue costs
// fit the 64-cycle gap):
//   the weight fragments of the next pair (ring of 4; positions 16, 17 = 0, 1 of the next step),
//   the row pass (V = T B) of the next pair,
//   the next step's patch -- raw reads in pair 8 (after the mid-step barrier that publishes it), then
//   its column pass T = B^T d row by row into tc as soon as the row of the current step is dead
//   (rows 0, 1 in pair 10, row 2 in pair 12, row 3 in pair 14),
//   and, in pairs 8..14, three of the 11 LDS-DMA pieces of the step after next.
#define SGV3D_WINO_MFMA(P, C, V, BF) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(V.C, BF.C, acc[P], 0, 0, 0)
template <int P>
__device__ __forceinline__ void wino_pair(f32x16 (&acc)[16], float4 (&tc)[4][4], float4 (&raw)[4][4], float4 &vc0,
                                          float4 &vc1, float4 &vn0, float4 &vn1, float4 (&bf)[4], const float4 *Bc,
                                          const float4 *Bn, const float4 *An, WinoLoader &ld) {
    static_assert((P & 1) == 0, "pairs start at even positions");
    constexpr int G = P >= 8 ? 3 * ((P - 8) / 2) : 100;   // first piece of this pair (none before the barrier)
    const float4 b0 = bf[P & 3], b1 = bf[(P + 1) & 3];
    SGV3D_WINO_MFMA(P, x, vc0, b0);
    SGV3D_SB();
    if constexpr (P + 2 < 16) bf[(P + 2) & 3] = Bc[(P + 2) * 128];
    else bf[(P + 2) & 3] = Bn[(P + 2 - 16) * 128];
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, x, vc1, b1);
    SGV3D_SB();
    if constexpr (P + 3 < 16) bf[(P + 3) & 3] = Bc[(P + 3) * 128];
    else bf[(P + 3) & 3] = Bn[(P + 3 - 16) * 128];
    if constexpr (G < 11) ld.template piece<(G < 11 ? G : 0)>();
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, y, vc0, b0);
    SGV3D_SB();
    {
        constexpr int Q = (P + 2) & 15;
        vn0 = wino_bt<(Q & 3)>(tc[Q >> 2][0], tc[Q >> 2][1], tc[Q >> 2][2], tc[Q >> 2][3]);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, y, vc1, b1);
    SGV3D_SB();
    {
        constexpr int Q = (P + 3) & 15;
        vn1 = wino_bt<(Q & 3)>(tc[Q >> 2][0], tc[Q >> 2][1], tc[Q >> 2][2], tc[Q >> 2][3]);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, z, vc0, b0);
    SGV3D_SB();
    if constexpr (P == 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = An[i * (2 * A_HALF) + (j & 1) * A_HALF + (j >> 1)];
    } else if constexpr (P == 10 || P == 12) {
        constexpr int R = P == 10 ? 0 : 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) tc[R][j] = wino_bt<R>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, z, vc1, b1);
    SGV3D_SB();
    if constexpr (P == 8) {
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = An[i * (2 * A_HALF) + (j & 1) * A_HALF + (j >> 1)];
    } else if constexpr (P == 10 || P == 14) {
        constexpr int R = P == 10 ? 1 : 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) tc[R][j] = wino_bt<R>(raw[0][j], raw[1][j], raw[2][j], raw[3][j]);
    }
    SGV3D_SB();
    SGV3D_WINO_MFMA(P, w, vc0, b0);
    SGV3D_SB();
    if constexpr (G + 1 < 11) ld.template piece<(G + 1 < 11 ? G + 1 : 0)>();
    SGV3D_SB();
    SGV3D_WINO_MFMA(P + 1, w, vc1, b1);
    SGV3D_SB();
    if constexpr (G + 2 < 11) ld.template piece<(G + 2 < 11 ? G + 2 : 0)>();
    SGV3D_SB();
}

__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];

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
    const int oy0 = by * 16, ox0 = bx * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, t = lane & 31;

    const int nsteps_all = a.cin / WK;
    const int kb = (int)((long long)nsteps_all * blockIdx.y / a.split_k);
    const int ke = (int)((long long)nsteps_all * (blockIdx.y + 1) / a.split_k);

    // ---- global sources of this lane's three patch slots and of the weight stream ----------------
    WinoLoader ld;
    ld.wave = wave;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = (wave * 3 + i) * 64 + lane;
        const int hh = s / A_PLANE, rem = s - hh * A_PLANE;
        const int row = rem / (2 * A_HALF), r2 = rem - row * (2 * A_HALF);
        const int par = r2 / A_HALF, ch = r2 - par * A_HALF;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + 2 * ch + par;
        const bool ok = (s < A_USED) & (ch < 9) & (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
        ld.asrc[i] = ok ? a.x + ((long long)(img * a.in_h + iy) * a.in_w + ix) * a.x_ld + a.x_coff + hh * 4 + kb * WK
                        : a.zeros;
        ld.ainc[i] = ok ? WK : 0;
    }
    ld.wsrc = a.w + ((size_t)tn * nsteps_all + kb) * (W_SLOTS * 4) + (wave * 8 * 64 + lane) * 4;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    const int abase = (h * A_ROWS + 2 * (wm * 4 + (t >> 3))) * (2 * A_HALF) + (t & 7);
    const int bbase = A_SLOTS + h * 64 + wn * 32 + t;

    // ---- prologue: steps kb (buffer 0) and kb+1 (buffer 1) in flight, first patch transformed ------
    ld.on = true;
    ld.dst = smem;
    ld.all();
    SGV3D_WINO_PUBLISH();
    ld.on = kb + 1 < ke;
    ld.dst = smem + BUF_SLOTS;
    ld.all();
    float4 tc[4][4], raw[4][4], bf[4], va0, va1, vb0, vb1;
    {
        const float4 *const A = smem + abase;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[i][j] = A[i * (2 * A_HALF) + (j & 1) * A_HALF + (j >> 1)];
        bf[0] = smem[bbase];
        bf[1] = smem[bbase + 128];
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

    // ---- main loop: three LDS buffers, one barrier per step (in its middle) ------------------------
    // Step s reads buffer s%3.  In the middle of step s: every wave retires its copies of step s+1
    // (vmcnt), the barrier publishes buffer (s+1)%3 and proves that buffer (s+2)%3 (last read in step
    // s-1) is free, and the copies of step s+2 start.  The second half of step s already reads the
    // next step's patch and runs its column pass, so MFMAs never wait at a step boundary.  (In the last
    // step those reads hit a stale but valid buffer and their results are dropped.)
    int slot = 0;
    for (int s = kb; s < ke; ++s) {
        const int nslot = slot == 2 ? 0 : slot + 1;
        const float4 *const Bc = smem + slot * BUF_SLOTS + bbase;
        const float4 *const Bn = smem + nslot * BUF_SLOTS + bbase;
        const float4 *const An = smem + nslot * BUF_SLOTS + abase;
        wino_pair<0>(acc, tc, raw, va0, va1, vb0, vb1, bf, Bc, Bn, An, ld);
        wino_pair<2>(acc, tc, raw, vb0, vb1, va0, va1, bf, Bc, Bn, An, ld);
        wino_pair<4>(acc, tc, raw, va0, va1, vb0, vb1, bf, Bc, Bn, An, ld);
        wino_pair<6>(acc, tc, raw, vb0, vb1, va0, va1, bf, Bc, Bn, An, ld);
        if (s + 1 < ke) SGV3D_WINO_PUBLISH();
        ld.on = s + 2 < ke;
        ld.dst = smem + (nslot == 2 ? 0 : nslot + 1) * BUF_SLOTS;
        SGV3D_SB();
        wino_pair<8>(acc, tc, raw, va0, va1, vb0, vb1, bf, Bc, Bn, An, ld);
        wino_pair<10>(acc, tc, raw, vb0, vb1, va0, va1, bf, Bc, Bn, An, ld);
        wino_pair<12>(acc, tc, raw, va0, va1, vb0, vb1, bf, Bc, Bn, An, ld);
        wino_pair<14>(acc, tc, raw, vb0, vb1, va0, va1, bf, Bc, Bn, An, ld);
        slot = nslot;
    }

    // ---- output transform + epilogue ---------------------------------------------------------------
    float *ws = a.split_k > 1 ? a.ws + (size_t)blockIdx.y * a.M * a.N : nullptr;
    const int col = tn * 64 + wn * 32 + t;
    if (col >= a.N) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tw = 8 * (e >> 2) + 4 * h + (e & 3);
        const int ty = wm * 4 + (tw >> 3), tx = tw & 7;
        const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
        // rows of A^T M:  r0 = m0 + m1 + m2,  r1 = m1 - m2 - m3   (per Winograd column j)
        float r0[4], r1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[j][e], m1 = acc[4 + j][e], m2 = acc[8 + j][e], m3 = acc[12 + j][e];
            r0[j] = m0 + m1 + m2;
            r1[j] = m1 - m2 - m3;
        }
        const float y00 = r0[0] + r0[1] + r0[2], y01 = r0[1] - r0[2] - r0[3];
        const float y10 = r1[0] + r1[1] + r1[2], y11 = r1[1] - r1[2] - r1[3];
        const float yv[2][2] = {{y00, y01}, {y10, y11}};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                if (oy + dy < a.out_h && ox + dx < a.out_w) {
                    const int row = (img * a.out_h + oy + dy) * a.out_w + ox + dx;
                    if (ws) ws[(size_t)row * a.N + col] = yv[dy][dx];
                    else conv_epilogue_store(a, row, col, yv[dy][dx]);
                }
            }
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
    return (size_t)((cout + 63) / 64) * (cin_pad / WK) * (W_SLOTS * 4);
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
    SGV3D_REQUIRE((long long)d->batch * d->in_h * d->in_w * d->x_ld < (1LL << 40), "conv2d_winograd_forward: input too large");
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
    a.wb_y = cdiv(d->out_h, 16);
    a.wb_x = cdiv(d->out_w, 16);
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
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kWinoLds) != hipSuccess)
            return fail(SGV3D_ELAUNCH, "conv2d_winograd_forward: cannot raise the dynamic LDS limit to %d", kWinoLds);
        attr_set = true;
    }
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(conv_wino_kernel, dim3(a.tiles_m * a.tiles_n, a.split_k), dim3(256), kWinoLds, st, a);
    if (a.split_k > 1) return launch_splitk_reduce(a, st);
    return check_launch("conv_wino_kernel");
}
