// MFMA implicit-GEMM convolution for gfx950, fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32:
// exact f32 FMA chain, 157 TFLOP/s dense peak), NHWC activations.
//
// Replaces the cuDNN convolutions PyTorch dispatches for every nn.Conv2d / nn.ConvTranspose2d on the
// reference's forward path (mmdet ResNet, mmdet3d SECONDFPN, HeightNet, BEV trunk, CenterHead:
// layers/backbones/lss_fpn.py:18-250,296-301; layers/heads/bev_height_head.py:75-110), with the
// BatchNorm (eval) scale/shift, residual add, ReLU and SE gate folded into the epilogue.
//
// GEMM view:  C[m][n] = sum_k A[m][k] * W[n][k]
//   m = (image, oh, ow) output pixel, n = output channel, k = (kh, kw, ci) with ci fastest, so a
//   16-byte chunk of A (4 consecutive k) is 4 consecutive channels of ONE input pixel in NHWC and is
//   fetched with one global_load_dwordx4; W is pre-packed [n][k] so both LDS tiles are k-contiguous.
//
// Workgroup: 256 threads = 4 waves (2 x 2), tile (64*WTM) x (64*WTN) x 32; each wave owns WTM x WTN
// MFMA tiles of 32x32 (16 accumulator VGPRs each).  LDS rows are padded to 36 floats (144 B): the
// 16 lanes of every ds_read_b128 service group then fall on 16 distinct 16-B slots (9*r mod 16 is a
// bijection), i.e. conflict-free fragment reads.  One ds_read_b128 per lane feeds FOUR MFMAs: lane
// half h supplies k = 8q + 4h + j for MFMA j of k-group q in both operands (the k order inside a
// group is permuted consistently for A and W, which only reorders the fp32 summation).
// Global loads for tile t+1 are issued before the MFMAs of tile t and written to the other LDS
// buffer afterwards (one barrier per k-tile); 2 workgroups/CU cover each other's barrier stalls.
// Block ids are remapped so that the workgroups sharing an XCD (private 4 MiB L2) work on
// consecutive m-tiles of the same n-tile, i.e. share one packed-weight panel.
#include "conv_common.hpp"

#include <stdlib.h>

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int kThreads = 256;

// 16 zero bytes that padded / out-of-image lanes load from
__device__ __attribute__((aligned(16))) float4 g_zero16 = {0.f, 0.f, 0.f, 0.f};

// tools/igemm_stamps.py builds a private copy of the library with -DSGV3D_IGEMM_STAMPS: every 97th workgroup of the f32
// kernel writes 4 cycle-counter stamps (entry, operands of the first k-tile in LDS, k loop done, epilogue issued)
#ifdef SGV3D_IGEMM_STAMPS
__device__ long long *g_igemm_dbg = nullptr;
#define IGEMM_STAMP(i) do { if (g_igemm_dbg && blockIdx.y == 0 && blockIdx.x % 97 == 0 && blockIdx.x / 97 < 64 && threadIdx.x == 0) g_igemm_dbg[(blockIdx.x / 97) * 4 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define IGEMM_STAMP(i) do { } while (0)
#endif

// Split-K second stage, row-linear outputs (NHWC with channel offset, f32, no SE gate, N % 4 == 0): four channels per
// thread, 16-byte accesses, 32-bit index arithmetic -- the same sums in the same order as the generic kernel below, at a
// fifth of its vector instructions (which are MFMA time of the SIMD they run on).
__global__ __launch_bounds__(256) void conv_splitk_reduce4_kernel(const ConvArgs a) {
    const unsigned n4 = (unsigned)a.N >> 2;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned total4 = (unsigned)a.M * n4;
    if (i >= total4) return;
    const unsigned row = i / n4, col = (i - row * n4) * 4u;
    const size_t stride = (size_t)a.M * a.N;
    const float *p = a.ws + (size_t)row * a.N + col;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < a.split_k; ++s) {
        const float4 t = *reinterpret_cast<const float4 *>(p + (size_t)s * stride);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 sc = a.scale ? *reinterpret_cast<const float4 *>(a.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = a.bias ? *reinterpret_cast<const float4 *>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (a.res) {
        const float4 r = *reinterpret_cast<const float4 *>(a.res + (size_t)row * a.res_ld + col);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4 *>(a.y + (size_t)row * a.y_ld + a.y_coff + col) = v;
}

// Split-K second stage: sums the split partials in fixed order and runs the common epilogue.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.M * a.N;
    if (i >= total) return;
    const int row = (int)(i / a.N), col = (int)(i - (long long)row * a.N);
    float v = 0.f;
    for (int s = 0; s < a.split_k; ++s) v += a.ws[(size_t)s * total + i];
    conv_epilogue_store(a, row, col, v);
}

// 128x128 needs ~280 VGPRs with the two register stages: it runs one workgroup per CU (1 wave/SIMD,
// the pipeline covers its own latency); the smaller tiles keep >= 2 waves/SIMD.
//
// SWAP (64x64 tile, NHWC output, no split-K, no SE gate; chosen by the launcher): the two MFMA operands trade places
// (C^T = W . X^T), which puts the PIXEL on the lane and 4 consecutive output channels in registers 4q..4q+3 of an
// accumulator tile.  Every product and the order of the k sum are the same, so results are bitwise those of the unswapped
// kernel -- but the epilogue moves 16 bytes per instruction: folded-BN terms, residual and output of a lane's 16 values are
// 4 + 4 + 4 + 4 dwordx4 accesses instead of 16 + 16 dword ones.  Meant for the expanding 1x1 convolutions of the ResNet
// bottlenecks (K = 64..512, 2 k-tiles of MFMA work against 32 KB of residual + output per workgroup); measured, it does
// not pay (see swap_epilogue_enabled below), so it is opt-in.
//
// PW (pointwise; chosen by the launcher for 1x1 / stride 1 / unpadded layers with cin % 32 == 0 -- the ResNet bottleneck
// convolutions and the grouped GEMM of the F(4x4) Winograd path): f32 MFMAs and vector-ALU instructions share the SIMD's lanes
// on this chip EVEN ACROSS WAVES (tools/ubench/mfma_valu_overlap.hip: a pure MFMA wave and a pure v_fma wave on one SIMD take
// the sum of their times, 933 + 584 -> 1445 us), so every vector instruction of the address arithmetic is MFMA time of the
// whole SIMD.  Here the A operand needs none inside the loop: the lane offset is fixed (out of range for rows past M), the
// k-tile offset is a scalar register, and past the last tile the last one is re-read (its data is never stored).
//
// NARROW (with WTM = WTN = 1): a 32 x 128 tile -- the four waves side by side along N, each one 32 x 32 MFMA tile -- for GEMMs
// whose row count is a multiple of 32 but not of 64: the grouped GEMM of the F(4x4) path pads the tiles of a position to the
// m-tile height, and 336 tiles are 352 rows instead of 384, 84 are 96 instead of 128.
//
// OCC (64 x 64 tile, FAST, not SWAP; desc.tile | SGV3D_TILE_OCC5): FIVE workgroups per CU instead of four -- for the small-K
// layers whose workgroups spend most of their life in the prologue / epilogue (2-4 k MFMA cycles behind 6-13 k cycles of memory
// latency), a fifth resident workgroup is a quarter more MFMA work to fill those cycles with.  The budget: exactly 32 KB of
// LDS (the 4-float row padding becomes an XOR swizzle of the 16-byte chunk with (row >> 1) & 7: conflict-free for the 16-lane
// phases of both the b128 stores and the b128 fragment reads) and <= 96 registers (one register stage instead of two: tile
// t+1 is requested at the top of tile t's phase and written to the other buffer just before the phase's barrier; the residual
// is read in the epilogue instead of being prefetched).  One more candidate of the per-layer measurement.
template <int WTM, int WTN, bool FAST, bool SWAP = false, bool PW = false, bool NARROW = false, bool OCC = false>
__global__ __launch_bounds__(kThreads, OCC ? 5 : (WTM * WTN == 4) ? 1 : 2) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(!OCC || (WTM == 1 && WTN == 1 && FAST && !SWAP && !NARROW), "the five-per-CU form is the plain 64x64 FAST tile");
    constexpr int LDX = OCC ? BK : LDK;            // floats per LDS row
    static_assert(!SWAP || (WTM == 1 && WTN == 1), "the swapped epilogue is written for the 64x64 tile");
    static_assert(!PW || FAST, "the pointwise specialisation is for the channel-chunk-major k order");
    static_assert(!NARROW || (WTM == 1 && WTN == 1 && !SWAP), "the narrow tile is one MFMA tile per wave");
    constexpr int BM = NARROW ? 32 : 64 * WTM, BN = NARROW ? 128 : 64 * WTN;
    constexpr int WSM = NARROW ? 32 : BM / 2, WSN = NARROW ? 32 : BN / 2;      // rows / columns between neighbouring waves
    constexpr int A_CH = BM / 32, B_CH = BN / 32;  // 16-B chunks per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    IGEMM_STAMP(0);
    float *const As0 = smem;
    float *const Bs0 = smem + BM * LDX;
    constexpr int kBufStride = (BM + BN) * LDX;
    // Tap-major K (cin % 32 != 0: the 7x7 stems): decoding k -> (tap, channel) takes two integer divisions
    // per thread and k-tile -- vector instructions that cost MFMA time on this chip.  They are done once per
    // workgroup instead: a table in LDS behind the tiles, one entry per 16-byte k-chunk:
    // {input offset of the chunk relative to the pixel, dy << 16 | dx}, dy = -1 for chunks past K.
    int2 *const klut = reinterpret_cast<int2 *>(smem + 2 * kBufStride);
    if constexpr (!FAST) {
        for (int kc = threadIdx.x; kc < a.k_pad / 4; kc += kThreads) {
            const int k = kc * 4;
            int2 e = make_int2(0, -1);
            if (k < a.K) {
                const int tap = k / a.cin, ci = k - tap * a.cin;
                const int kh = tap / a.kw, dy = kh * a.dil, dx = (tap - kh * a.kw) * a.dil;
                e = make_int2((dy * a.in_w + dx) * a.x_ld + ci, (dy << 16) | dx);
            }
            klut[kc] = e;
        }
        __syncthreads();
    }

    // ---- XCD-aware tile mapping (bijective for any tile count) --------------------------------
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // Default walk: the output-channel tile changes slowest (consecutive workgroups share a weight tile).  korder bit 1
    // (desc.tile | SGV3D_TILE_MFIRST): the m-tile changes slowest -- its input rows are fetched from HBM once instead of once
    // per channel tile (tools/layer_traffic.py); pays on the bandwidth-heavy layers (64->256 at 216x384, 2560->512), costs on
    // the others, so it is one more candidate of the first-call measurement.
    int tn, tm;
    if (a.korder & 2) {
        tm = (int)((unsigned)logical / (unsigned)a.tiles_n);
        tn = logical - tm * a.tiles_n;
    } else {
        tn = (int)((unsigned)logical / (unsigned)a.tiles_m);     // (unsigned: half the scalar instructions of a signed division)
        tm = logical - tn * a.tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int cc = tid & 7;    // 16-B chunk column inside the 32-wide k-tile
    const int r0 = tid >> 3;   // first row handled by this thread (then +32, +64, ...)
    const int cs = OCC ? (cc ^ ((r0 >> 1) & 7)) : cc;      // chunk slot inside the LDS row (rows +32 i share the key)

    // ---- per-thread A rows -----------------------------------------------------------------
    // Operands are read with buffer loads: resource (base, size) and the uniform part of the address in
    // scalar registers, a 32-bit byte offset per lane -- out-of-image / padding lanes carry an out-of-range
    // offset and read zeros.  fp32 MFMAs and VALU instructions share the SIMD's lanes on this chip, so
    // every vector instruction saved in the loop (64-bit pointer arithmetic, selects) is MFMA time.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_none = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, 0, 0x00020000);   // num_records 0: reads zeros
    // (grouped GEMM, conv_gemm_grouped: the m-tile's group picks its weight block; wave-uniform)
    const float *const w_grp = a.wb_y > 0 ? a.w + (size_t)((unsigned)m0 / (unsigned)a.wb_y) * (size_t)a.wb_x : a.w;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w_grp, 0, (int)a.w_bytes, 0x00020000);
    // The prologue runs in the shadow of the other workgroups' MFMAs: at four waves per SIMD an instruction of this wave issues
    // every 8-13 cycles, and ~1000 scalar + vector instructions of address set-up were 9-13 k cycles before the first k-tile
    // reached LDS (tools/igemm_stamps.py) -- as long as the whole k loop of a 256-deep 1x1 layer.  Hence the shortcuts below.
    const bool pointwise = PW || (a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.in_h == a.m_h && a.in_w == a.m_w);
    unsigned a_off[A_CH];
    int a_ih0[A_CH], a_iw0[A_CH];
    bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        if constexpr (PW) {          // rows past M carry an out-of-range lane offset for good
            a_ih0[i] = 0;
            a_iw0[i] = 0;
            a_off[i] = a_ok[i] ? (unsigned)(((long long)mm * a.x_ld + a.x_coff + cc * 4) * 4) : 0xffffffffu;
            continue;
        }
        if (pointwise) {             // 1x1 / stride 1 / no padding: the input pixel IS the output pixel, no division
            a_ih0[i] = 0;
            a_iw0[i] = 0;
            a_off[i] = (unsigned)(((long long)mm * a.x_ld + a.x_coff + cc * 4) * 4);
            continue;
        }
        const int t = (int)((unsigned)mm / (unsigned)a.m_w);
        const int ow = mm - t * a.m_w;
        const int n = (int)((unsigned)t / (unsigned)a.m_h);
        const int oh = t - n * a.m_h;
        a_ih0[i] = oh * a.stride - a.pad;
        a_iw0[i] = ow * a.stride - a.pad;
        // (wraps for rows / columns in the padding; only used when the tap is inside the image)
        a_off[i] = (unsigned)((((long long)(n * a.in_h + a_ih0[i]) * a.in_w + a_iw0[i]) * a.x_ld + a.x_coff + cc * 4) * 4);
    }
    unsigned b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) b_off[i] = (unsigned)(((size_t)(n0 + r0 + 32 * i) * a.k_pad + cc * 4) * 4);

    // Two register stages: tile t+2 is fetched while tile t is multiplied and tile t+1 is written to
    // the other LDS buffer between the two MFMA halves.
    float4 ra0[A_CH], rb0[B_CH], ra1[A_CH], rb1[B_CH];
    // k state of the NEXT tile to fetch.  FAST (cin % 32 == 0, weights packed channel-chunk-major):
    // a k-tile is (32-channel chunk c0, tap (kh, kw)) -- wave-uniform, advanced incrementally, and the
    // nine taps of one chunk are fetched back to back so their overlapping input pixels hit in L1/L2.
    // General path: k = tap * cin + ci, decoded per thread with integer division.
    // split-K: blockIdx.y owns k-tiles [kt_begin, nkt) of a balanced partition
    // FAST path only: kernel rows whose input row lies outside the image for EVERY output row of this m-tile
    // are skipped altogether (the dilated ASPP convolutions at 54x96 with dilation 12 / 18 lose up to a third
    // of their taps this way; the zero-block loads they would have made feed MFMAs with zeros).  The valid
    // rows form a contiguous range [kh_lo, kh_hi]; k-tiles are counted over the valid list.
    int kh_lo = 0, kh_hi = a.kh - 1;
    if constexpr (FAST && !PW) {
        if (a.mode != SGV3D_CONV_DECONV && a.kh > 1) {
            const int hw = a.m_h * a.m_w;
            const int m_last = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
            const int img0 = m0 / hw, img1 = m_last / hw;
            if (img0 == img1) {
                const int r_first = (m0 - img0 * hw) / a.m_w, r_last = (m_last - img0 * hw) / a.m_w;
                const int lo_num = a.pad - r_last * a.stride;                 // kh * dil >= lo_num
                const int hi_num = a.in_h - 1 + a.pad - r_first * a.stride;   // kh * dil <= hi_num
                const int lo = lo_num <= 0 ? 0 : (lo_num + a.dil - 1) / a.dil;
                const int hi = hi_num < 0 ? -1 : hi_num / a.dil;
                if (lo <= hi && lo < a.kh) {
                    kh_lo = lo;
                    kh_hi = hi < a.kh - 1 ? hi : a.kh - 1;
                }
            }
        }
    }
    const int nkh = kh_hi - kh_lo + 1;
    const int nkt_all = FAST ? (a.cin / BK) * nkh * a.kw : a.k_pad / BK;
    // (nkt_all <= 65536 and split_k <= 64: 32-bit unsigned arithmetic; no division at all without split-K)
    int kt_begin = 0, nkt = nkt_all;                                   // [kt_begin, nkt): this workgroup's k-tiles
    if (a.split_k > 1) {
        kt_begin = (int)((unsigned)nkt_all * blockIdx.y / (unsigned)a.split_k);
        nkt = (int)((unsigned)nkt_all * (blockIdx.y + 1) / (unsigned)a.split_k);
    }
    int ld_kt = kt_begin, ld_kh = kh_lo, ld_kw = 0, ld_c0 = 0, ld_kp = kh_lo * a.kw;
    if constexpr (FAST) {
        if (kt_begin != 0) {
            const unsigned per_chunk = nkh * a.kw;
            const unsigned kt0 = kt_begin < nkt_all ? kt_begin : nkt_all - 1;     // (an empty split slice reads the last tile)
            const unsigned chunk = kt0 / per_chunk, rem = kt0 - chunk * per_chunk;
            ld_c0 = chunk * BK;
            ld_kh = kh_lo + rem / (unsigned)a.kw;
            ld_kw = rem - (rem / (unsigned)a.kw) * a.kw;
            ld_kp = chunk * a.kh * a.kw + ld_kh * a.kw + ld_kw;     // k-tile index in the packed weight order
        }
    }

    // Loads are unconditional: lanes whose tap falls outside the image (or whose row / k is padding)
    // read a 16-byte block of zeros, so the stage is straight-line global_load_dwordx4 with no
    // exec-mask branches and no post-load selects.
#define SGV3D_LOAD_TILE(RA, RB)                                                                       \
    do {                                                                                              \
        int dy_, dx_, koff_;                                                                          \
        /* past the last tile (pipeline drain) nothing is fetched for A and the last B tile is */     \
        /* re-read: keeps the stage branch-free so the compiler can count vmcnt exactly        */     \
        bool kvalid_ = ld_kt < nkt;                                                                   \
        const int ktb_ = FAST ? ld_kp : (ld_kt < nkt ? ld_kt : nkt - 1);                              \
        if constexpr (FAST) {                                                                         \
            dy_ = ld_kh * a.dil;                                                                      \
            dx_ = ld_kw * a.dil;                                                                      \
            koff_ = (dy_ * a.in_w + dx_) * a.x_ld + ld_c0;                                            \
        } else {                                                                                      \
            const int2 e_ = klut[ktb_ * (BK / 4) + cc];                                               \
            kvalid_ = kvalid_ & (e_.y >= 0);                                                          \
            dy_ = e_.y >> 16;                                                                         \
            dx_ = e_.y & 0xffff;                                                                      \
            koff_ = e_.x - cc * 4;       /* a_off already holds this thread's chunk column */          \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                            \
            if constexpr (PW) {   /* fixed lane offset + scalar channel-chunk offset: no vector instruction; */ \
                /* past the last tile the EMPTY resource (scalar select): nothing is fetched             */ \
                const f32x4n t_ = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(kvalid_ ? x_rsrc : x_none, a_off[i], ld_c0 * 4, 0)); \
                RA[i].x = t_.x; RA[i].y = t_.y; RA[i].z = t_.z; RA[i].w = t_.w;                       \
            } else {                                                                                  \
                const int ih_ = a_ih0[i] + dy_, iw_ = a_iw0[i] + dx_;                                 \
                const bool v_ = a_ok[i] & kvalid_ & ((unsigned)ih_ < (unsigned)a.in_h) &              \
                                ((unsigned)iw_ < (unsigned)a.in_w);                                   \
                const unsigned vo_ = v_ ? a_off[i] + (unsigned)(koff_ * 4) : 0xffffffffu;             \
                const f32x4n t_ = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo_, 0, 0)); \
                RA[i].x = t_.x; RA[i].y = t_.y; RA[i].z = t_.z; RA[i].w = t_.w;                       \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                                            \
            const f32x4n t_ = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_off[i], ktb_ * (BK * 4), 0)); \
            RB[i].x = t_.x; RB[i].y = t_.y; RB[i].z = t_.z; RB[i].w = t_.w;                           \
        }                                                                                             \
        ++ld_kt;                                                                                      \
        if constexpr (PW) {                                                                           \
            if (ld_kt < nkt) { ld_c0 += BK; ld_kp = ld_c0 / BK; }                                     \
        } else if constexpr (FAST) {                                                                  \
            if (ld_kt < nkt) { /* past the end the state stays on the last valid tile (drain re-reads it) */ \
                if (++ld_kw == a.kw) {                                                                \
                    ld_kw = 0;                                                                        \
                    if (++ld_kh > kh_hi) { ld_kh = kh_lo; ld_c0 += BK; }                              \
                }                                                                                     \
                ld_kp = (ld_c0 / BK) * a.kh * a.kw + ld_kh * a.kw + ld_kw;                            \
            }                                                                                         \
        }                                                                                             \
    } while (0)

#define SGV3D_STORE_TILE(RA, RB, BUF)                                                                 \
    do {                                                                                              \
        float *As_ = As0 + (BUF) * kBufStride, *Bs_ = Bs0 + (BUF) * kBufStride;                       \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                              \
            *reinterpret_cast<float4 *>(As_ + (r0 + 32 * i) * LDX + cs * 4) = RA[i];                  \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                              \
            *reinterpret_cast<float4 *>(Bs_ + (r0 + 32 * i) * LDX + cs * 4) = RB[i];                  \
    } while (0)

    // ---- MFMA fragments ------------------------------------------------------------------------
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = NARROW ? 0 : wave >> 1, wn = NARROW ? wave : wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int a_frag_off = (wm * WSM + lr) * LDX + (OCC ? 0 : lh * 4);
    const int b_frag_off = (wn * WSN + lr) * LDX + (OCC ? 0 : lh * 4);
    // offset of k-group KQ's 16-byte chunk (2 KQ + lh) inside the lane's row: plain, or swizzled with the row's key
    // ((row >> 1) & 7 = (lr >> 1) & 7 for every row of the lane: the wave / MFMA-tile offsets are multiples of 32)
    int fo[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) fo[kq] = OCC ? (((2 * kq + lh) ^ ((lr >> 1) & 7)) * 4) : kq * 8;

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    // Fragment double buffering at k-group granularity: the ds_read_b128 of k-group q+1 are issued
    // before the MFMAs of k-group q, and the first k-group of the NEXT tile is fetched right after the
    // barrier that publishes it, under the last MFMAs of the current tile -> no exposed LDS latency.
    float4 fa0[WTM], fb0[WTN], fa1[WTM], fb1[WTN];
#define SGV3D_READ_FRAG(FA, FB, BUF, KQ)                                                              \
    do {                                                                                              \
        const float *Aw_ = As0 + (BUF) * kBufStride + a_frag_off + fo[KQ];                            \
        const float *Bw_ = Bs0 + (BUF) * kBufStride + b_frag_off + fo[KQ];                            \
        _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt) {                                          \
            const float4 t_ = *reinterpret_cast<const float4 *>(Aw_ + mt * 32 * LDX);                 \
            FA[mt].x = t_.x; FA[mt].y = t_.y; FA[mt].z = t_.z; FA[mt].w = t_.w;                       \
        }                                                                                             \
        _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt) {                                          \
            const float4 t_ = *reinterpret_cast<const float4 *>(Bw_ + nt * 32 * LDX);                 \
            FB[nt].x = t_.x; FB[nt].y = t_.y; FB[nt].z = t_.z; FB[nt].w = t_.w;                       \
        }                                                                                             \
    } while (0)
#define SGV3D_MFMA_KQ(FA, FB)                                                                         \
    do {                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt) {                                      \
                const float av = j == 0 ? FA[mt].x : j == 1 ? FA[mt].y : j == 2 ? FA[mt].z : FA[mt].w; \
                _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt) {                                  \
                    const float bv = j == 0 ? FB[nt].x : j == 1 ? FB[nt].y : j == 2 ? FB[nt].z : FB[nt].w; \
                    acc[mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[mt][nt], 0, 0, 0)  \
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt][nt], 0, 0, 0); \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define SGV3D_SB() __builtin_amdgcn_sched_barrier(0)

    // One k-tile held in LDS buffer BUF: fetch tile +2 into (RA, RB), publish tile +1 from (SA, SB).
#define SGV3D_PHASE(BUF, RA, RB, SA, SB, HAVE_NEXT)                                                   \
    do {                                                                                              \
        SGV3D_LOAD_TILE(RA, RB);                                                                      \
        SGV3D_READ_FRAG(fa1, fb1, BUF, 1);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa0, fb0);                                                                      \
        SGV3D_SB();                                                                                   \
        SGV3D_READ_FRAG(fa0, fb0, BUF, 2);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa1, fb1);                                                                      \
        SGV3D_SB();                                                                                   \
        if (HAVE_NEXT) SGV3D_STORE_TILE(SA, SB, (BUF) ^ 1);                                           \
        SGV3D_READ_FRAG(fa1, fb1, BUF, 3);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa0, fb0);                                                                      \
        SGV3D_SB();                                                                                   \
        __syncthreads();                                                                              \
        if (HAVE_NEXT) SGV3D_READ_FRAG(fa0, fb0, (BUF) ^ 1, 0);                                       \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa1, fb1);                                                                      \
        SGV3D_SB();                                                                                   \
    } while (0)

    // OCC: one register stage.  Tile t+1 is requested at the top of tile t's phase and published just before the barrier.
#define SGV3D_PHASE_OCC(BUF, HAVE_NEXT)                                                               \
    do {                                                                                              \
        if constexpr (HAVE_NEXT) SGV3D_LOAD_TILE(ra0, rb0);                                           \
        else SGV3D_PREFETCH_RES();                                                                    \
        SGV3D_READ_FRAG(fa1, fb1, BUF, 1);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa0, fb0);                                                                      \
        SGV3D_SB();                                                                                   \
        SGV3D_READ_FRAG(fa0, fb0, BUF, 2);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa1, fb1);                                                                      \
        SGV3D_SB();                                                                                   \
        SGV3D_READ_FRAG(fa1, fb1, BUF, 3);                                                            \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa0, fb0);                                                                      \
        SGV3D_SB();                                                                                   \
        if constexpr (HAVE_NEXT) {                                                                    \
            SGV3D_STORE_TILE(ra0, rb0, (BUF) ^ 1);                                                    \
            __syncthreads();                                                                          \
            SGV3D_READ_FRAG(fa0, fb0, (BUF) ^ 1, 0);                                                  \
        }                                                                                             \
        SGV3D_SB();                                                                                   \
        SGV3D_MFMA_KQ(fa1, fb1);                                                                      \
        SGV3D_SB();                                                                                   \
    } while (0)

    // Residual prefetch (64x64 tile, fast-path epilogue): the small-K layers that carry a residual (the expanding 1x1
    // convolutions of the bottlenecks) spend a third of a workgroup's life waiting for the residual rows they only
    // ask for after the k loop; asked for here, the 16 values per lane arrive under the loop.
    constexpr bool kPrefetchRes = WTM * WTN <= 2;
    // (GROUP_PLANES -- the per-branch hidden maps of the two-kernel head path -- is row-linear too when a wave's columns
    // stay inside one group: plane base + row * group width)
    const bool fast_epi = (a.split_k > 1 || (a.mode == SGV3D_CONV_NORMAL && a.gate == nullptr) ||
                           (a.mode == SGV3D_CONV_GROUP_PLANES && a.gate == nullptr && a.ks % WSN == 0)) && m0 + BM <= a.M;
    // the first two k-tiles are asked for BEFORE the residual rows: memory returns in order, and the wait for k-tile 0
    // (the workgroup's prologue: 9-13 k cycles on a loaded chip, tools/igemm_stamps.py) must not queue behind 16 more loads
    SGV3D_LOAD_TILE(ra0, rb0);                       // tile 0
    if constexpr (!OCC) SGV3D_LOAD_TILE(ra1, rb1);   // tile 1
    float resv[WTM][WTN][16];
    // SWAP: the lane's pixel row (m0 + wm * 32 + lr) and its 4 channel quads (n0 + wn * 32 + 8 q + 4 lh): residual, folded-BN
    // scale and shift are 4 dwordx4 loads each, asked for here so that they arrive under the k loop
    f32x4n sw_res[4];
    if constexpr (SWAP) {
        const int wmu_ = __builtin_amdgcn_readfirstlane(wm);
        const long long tile_row = m0 + wmu_ * 32;
        const bool rok = tile_row + lr < a.M;
        const bool has_res = a.res != nullptr;
        const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(has_res ? a.res + tile_row * a.res_ld : a.zeros), 0, has_res ? (int)0xffffff00u : 0, 0x00020000);
        const int c0 = n0 + wn * 32 + 4 * lh;
        const unsigned roff = (rok && c0 < a.N) ? ((unsigned)lr * (unsigned)a.res_ld + (unsigned)c0) * 4u : 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q)      // (quad q: channels c0 + 8 q; past N the offset runs out of the resource: zeros)
            sw_res[q] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(
                pr_rsrc, (has_res && c0 + 8 * q < a.N) ? roff : 0xffffffffu, 32 * q, 0));
    }
    // (OCC: the same loads are issued at the top of the LAST k-tile's phase instead -- the staging registers of the k loop are
    //  free by then, and one k-tile of MFMAs still covers their latency)
#define SGV3D_PREFETCH_RES()                                                                          \
    do {                                                                                              \
        const bool want = fast_epi && a.split_k <= 1 && a.res != nullptr;                             \
        const long long tile_row = m0 + __builtin_amdgcn_readfirstlane(wm) * WSM;                     \
        const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(                     \
            (void *)(want ? a.res + tile_row * a.res_ld : a.zeros), 0, want ? (int)0xffffff00u : 0, 0x00020000); \
        _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt) {                                          \
            const int col = n0 + wn * WSN + nt * 32 + lr;                                             \
            const unsigned roff0 = (want && col < a.N) ? (4u * lh * (unsigned)a.res_ld + col) * 4u : 0xffffffffu; \
            _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt)                                        \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                      \
                    const unsigned r_ = mt * 32 + (e & 3) + 8 * (e >> 2);                             \
                    resv[mt][nt][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr_rsrc, roff0, r_ * a.res_ld * 4u, 0)); \
                }                                                                                     \
        }                                                                                             \
    } while (0)
    if constexpr (kPrefetchRes && !SWAP && !OCC) SGV3D_PREFETCH_RES();
    SGV3D_STORE_TILE(ra0, rb0, 0);
    __syncthreads();
    IGEMM_STAMP(1);
    SGV3D_READ_FRAG(fa0, fb0, 0, 0);
    if constexpr (OCC) {
        // every phase but the last in pairs (buffer 0, buffer 1); the last one is peeled: it asks for the residual rows instead
        // of a next tile, into the registers the staging no longer needs
        int kt = kt_begin;
        for (; kt + 2 < nkt; kt += 2) {
            SGV3D_PHASE_OCC(0, true);
            SGV3D_PHASE_OCC(1, true);
        }
        if (kt + 1 < nkt) {
            SGV3D_PHASE_OCC(0, true);
            SGV3D_PHASE_OCC(1, false);
        } else if (kt < nkt) {
            SGV3D_PHASE_OCC(0, false);
        }
    } else {
        for (int kt = kt_begin; kt < nkt; kt += 2) {
            SGV3D_PHASE(0, ra0, rb0, ra1, rb1, kt + 1 < nkt);      // tile kt in buffer 0
            if (kt + 1 >= nkt) break;
            SGV3D_PHASE(1, ra1, rb1, ra0, rb0, kt + 2 < nkt);      // tile kt+1 in buffer 1
        }
    }
    IGEMM_STAMP(2);
#undef SGV3D_LOAD_TILE
#undef SGV3D_STORE_TILE
#undef SGV3D_READ_FRAG
#undef SGV3D_MFMA_KQ
#undef SGV3D_PHASE
#undef SGV3D_PHASE_OCC
#undef SGV3D_PREFETCH_RES
#undef SGV3D_SB

    // ---- epilogue ------------------------------------------------------------------------------------
    // Fast path (row-linear layouts: NHWC output with channel offset, or the split-K partials; whole
    // m-tile inside M; no SE gate): the small-K layers (ResNet 1x1 convolutions) are bound by the
    // instructions of this epilogue, so it is a handful per output -- every channel-only term hoisted,
    // buffer stores addressed as [uniform tile base in the resource] + [per-lane VGPR: channel and the
    // lane half's 4 rows, or out of range for padded channels] + [scalar: row inside the wave's tile].
    if constexpr (SWAP) {
        const int wmu = __builtin_amdgcn_readfirstlane(wm);
        const long long tile_row = m0 + wmu * 32;
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.y + tile_row * a.y_ld + a.y_coff), 0, (int)0xffffff00u, 0x00020000);
        const float floor_ = a.relu ? 0.f : -__builtin_inff();
        const __amdgpu_buffer_rsrc_t sc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.scale ? a.scale : a.zeros), 0, a.scale ? a.N * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t sh_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.bias ? a.bias : a.zeros), 0, a.bias ? a.N * 4 : 0, 0x00020000);
        const bool no_scale = a.scale == nullptr;                 // (an absent scale reads zeros: means 1)
        const int c0 = n0 + wn * 32 + 4 * lh;
        const bool rok = tile_row + lr < a.M;
        const unsigned yoff = rok ? ((unsigned)lr * (unsigned)a.y_ld + (unsigned)c0) * 4u : 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // folded-BN terms of the quad: L2 / scalar-cache resident, asked for one quad ahead by the compiler's scheduling
            const f32x4n sc4 = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc, (unsigned)c0 * 4u, 32 * q, 0));
            const f32x4n sh4 = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(sh_rsrc, (unsigned)c0 * 4u, 32 * q, 0));
            f32x4n o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sc_ = no_scale ? 1.f : sc4[j];
                float v_ = acc[0][0][4 * q + j] * sc_ + sh4[j];
                v_ += sw_res[q][j];
                o[j] = fmaxf(v_, floor_);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, y_rsrc, (c0 + 8 * q < a.N) ? yoff : 0xffffffffu, 32 * q, 0);
        }
        IGEMM_STAMP(3);
        return;
    }
    if (fast_epi) {
        const bool partial = a.split_k > 1;
        const int wmu = __builtin_amdgcn_readfirstlane(wm);
        const bool planes = !partial && a.mode == SGV3D_CONV_GROUP_PLANES;
        const int grp = planes ? (n0 + __builtin_amdgcn_readfirstlane(wn) * WSN) / a.ks : 0;
        const int col_sub = grp * a.ks;                      // columns are counted inside the group's plane
        const unsigned ld = partial ? (unsigned)a.N : planes ? (unsigned)a.ks : (unsigned)a.y_ld;
        const long long tile_row = m0 + wmu * WSM;
        const float *const ybase = partial ? a.ws + ((size_t)blockIdx.y * a.M + tile_row) * a.N
                                   : planes ? a.y + ((size_t)grp * a.M + tile_row) * a.ks
                                            : a.y + tile_row * a.y_ld + a.y_coff;
        const __amdgpu_buffer_rsrc_t y_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)ybase, 0, (int)0xffffff00u, 0x00020000);
        const bool has_res = !partial && a.res != nullptr;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(has_res ? a.res + tile_row * a.res_ld : a.zeros), 0, has_res ? (int)0xffffff00u : 0, 0x00020000);
        const float floor_ = (!partial && a.relu) ? 0.f : -__builtin_inff();
        // raw: nothing to apply (split-K partials; the grouped GEMM of the F(4x4) path): the accumulators are stored as they are
        const bool raw = partial || (a.scale == nullptr && a.bias == nullptr && !has_res && !a.relu);
        unsigned voff[WTN], roff[WTN];
        float sc[WTN], sh[WTN];
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt) {
            const int col = n0 + wn * WSN + nt * 32 + lr;
            const bool ok = col < a.N;
            voff[nt] = ok ? (4u * lh * ld + (col - col_sub)) * 4u : 0xffffffffu;
            roff[nt] = ok ? (4u * lh * (unsigned)a.res_ld + col) * 4u : 0xffffffffu;
            sc[nt] = (!partial && ok && a.scale) ? a.scale[col] : 1.f;
            sh[nt] = (!partial && ok && a.bias) ? a.bias[col] : 0.f;
        }
#define SGV3D_EPI_F(MT, NT, E)                                                                        \
    {                                                                                                 \
        const unsigned r_ = (MT) * 32 + ((E) & 3) + 8 * ((E) >> 2);                                   \
        float v_ = acc[MT][NT][E];                                                                    \
        if (!raw) {                                                                                   \
            v_ = v_ * sc[NT] + sh[NT];                                                                \
            if constexpr (kPrefetchRes) {                                                             \
                v_ += resv[MT][NT][E]; /* zeros when there is no residual */                          \
            } else if (has_res)                                                                       \
                v_ += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, roff[NT], r_ * a.res_ld * 4u, 0)); \
            v_ = fmaxf(v_, floor_);                                                                   \
        }                                                                                             \
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), y_rsrc, voff[NT], r_ * ld * 4u, 0); \
    }
#define SGV3D_EPI_FT(MT, NT)                                                                          \
    SGV3D_EPI_F(MT, NT, 0) SGV3D_EPI_F(MT, NT, 1) SGV3D_EPI_F(MT, NT, 2) SGV3D_EPI_F(MT, NT, 3)       \
    SGV3D_EPI_F(MT, NT, 4) SGV3D_EPI_F(MT, NT, 5) SGV3D_EPI_F(MT, NT, 6) SGV3D_EPI_F(MT, NT, 7)       \
    SGV3D_EPI_F(MT, NT, 8) SGV3D_EPI_F(MT, NT, 9) SGV3D_EPI_F(MT, NT, 10) SGV3D_EPI_F(MT, NT, 11)     \
    SGV3D_EPI_F(MT, NT, 12) SGV3D_EPI_F(MT, NT, 13) SGV3D_EPI_F(MT, NT, 14) SGV3D_EPI_F(MT, NT, 15)
        SGV3D_EPI_FT(0, 0)
        if constexpr (WTN > 1) { SGV3D_EPI_FT(0, 1) }
        if constexpr (WTM > 1) {
            SGV3D_EPI_FT(1, 0)
            if constexpr (WTN > 1) { SGV3D_EPI_FT(1, 1) }
        }
#undef SGV3D_EPI_FT
#undef SGV3D_EPI_F
        IGEMM_STAMP(3);
        return;
    }
    // General path (pixel-shuffle / NCHW / grouped-plane layouts, SE gate, ragged last m-tile); expanded by
    // hand: the accumulators must keep compile-time register indices.
    float *ws = a.split_k > 1 ? a.ws + (size_t)blockIdx.y * a.M * a.N : nullptr;
    const int row_base = m0 + wm * WSM + 4 * lh;
    const int col_base = n0 + wn * WSN + lr;
#define SGV3D_EPI_E(MT, NT, E)                                                                        \
    {                                                                                                 \
        const int col_ = col_base + (NT) * 32;                                                        \
        const int row_ = row_base + (MT) * 32 + ((E) & 3) + 8 * ((E) >> 2);                           \
        if (col_ < a.N && row_ < a.M) {                                                               \
            if (ws) ws[(size_t)row_ * a.N + col_] = acc[MT][NT][E];                                   \
            else conv_epilogue_store(a, row_, col_, acc[MT][NT][E]);                                  \
        }                                                                                             \
    }
#define SGV3D_EPI_TILE(MT, NT)                                                                        \
    SGV3D_EPI_E(MT, NT, 0) SGV3D_EPI_E(MT, NT, 1) SGV3D_EPI_E(MT, NT, 2) SGV3D_EPI_E(MT, NT, 3)       \
    SGV3D_EPI_E(MT, NT, 4) SGV3D_EPI_E(MT, NT, 5) SGV3D_EPI_E(MT, NT, 6) SGV3D_EPI_E(MT, NT, 7)       \
    SGV3D_EPI_E(MT, NT, 8) SGV3D_EPI_E(MT, NT, 9) SGV3D_EPI_E(MT, NT, 10) SGV3D_EPI_E(MT, NT, 11)     \
    SGV3D_EPI_E(MT, NT, 12) SGV3D_EPI_E(MT, NT, 13) SGV3D_EPI_E(MT, NT, 14) SGV3D_EPI_E(MT, NT, 15)
    SGV3D_EPI_TILE(0, 0)
    if constexpr (WTN > 1) { SGV3D_EPI_TILE(0, 1) }
    if constexpr (WTM > 1) {
        SGV3D_EPI_TILE(1, 0)
        if constexpr (WTN > 1) { SGV3D_EPI_TILE(1, 1) }
    }
#undef SGV3D_EPI_TILE
#undef SGV3D_EPI_E
}

// ------------------------------------------------------------------------------------------------
// bf16 variant (BASELINE cfg-3 / cfg-5 ask for bf16 compute): same GEMM view, same fp32 operands in HBM, same
// fp32 accumulators and epilogue -- the operands are rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32)
// on their way from the load registers into LDS and multiplied by v_mfma_f32_32x32x16_bf16 (dense peak 2.5 PFLOP/s,
// 16x the fp32 rate), so the loop is bound by the operand traffic, not by MFMA.  LDS rows hold the 32 k of a tile as
// 64 bytes + 16 bytes of padding (row stride 80 B: the 16 lanes of a ds_read_b128 service group fall on 16 distinct
// 16-byte slots, 5 r mod 16 being a bijection); one ds_read_b128 per lane and operand feeds one MFMA of 16 k.
// Lane half h supplies k = 16 s + 8 h + j (j = 0..7) of k-step s for BOTH operands, so whatever k the hardware
// assigns to a lane's slot, A and W agree on it.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// Row stride in bf16 elements: the 32 k of a tile, 64 bytes, NO padding.  Bank conflicts are avoided by an XOR swizzle of the
// row's four 16-byte chunks with bits 2..3 of the row index: the 16 lanes of a ds_read_b128 service group (16 consecutive
// rows, same logical chunk) then touch 16 distinct 16-byte slots of the 256-byte bank row, and a half-wave of the 8-byte
// stores covers 4 consecutive rows = 256 contiguous bytes.  (An 80-byte padded row made the reads conflict-free but put
// every fourth row of the stores on the banks of the first: SQ_LDS_BANK_CONFLICT was a third of the LDS cycles.)
constexpr int LDKB = BK;

//
// SPLIT3 ("f32x3"): float32-accurate products on the bf16 matrix cores.  Every operand x is split exactly into three
// bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the remainders are exact in
// f32), kept as three planes in LDS, and a . b is summed from the six partial products whose weight is >= 2^-16 of the
// leading one: hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi, each exact in the f32 accumulator.  The dropped terms
// (mid.lo, lo.mid, lo.lo) are below 2^-23 relative: the error of a product is that of one f32 rounding.  Six bf16 MFMAs
// of K = 16 take 192 cycles where the f32 MFMA needs 512 for the same K.
//
// XB / YB (sgv3d_conv2d_forward_bf16io): bf16 ACTIVATIONS in HBM.  XB: the input tensor is bf16 -- a thread's 4 k of a tile
// are one 8-byte buffer load that goes to LDS as it is (no conversion, half the bytes).  YB: output (and residual) are
// bf16 -- the MFMA operands are swapped (C^T = W . X^T), which puts the PIXEL on the lane and 4 consecutive output
// channels in registers 4g..4g+3; each wave stages its 32 x 32 tile through LDS in fp32 and leaves it as 16-byte stores
// of 8 channels per lane (one 64-byte row of a pixel per 4 lanes), after folded BN, residual (bf16, added in fp32) and
// ReLU.  NORMAL mode only (concat offsets allowed, multiples of 8 channels).
template <int WTM, int WTN, bool FAST, bool SPLIT3, bool XB = false, bool YB = false>
__global__ __launch_bounds__(kThreads, (SPLIT3 && WTM * WTN >= 2) ? 1 : 2) void conv_igemm_bf16_kernel(const ConvArgs a) {
    static_assert(!(SPLIT3 && (XB || YB)), "f32x3 works on f32 tensors");
    constexpr int XES = XB ? 2 : 4;                            // bytes per input element
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    constexpr int A_CH = BM / 32, B_CH = BN / 32;
    constexpr int NP = SPLIT3 ? 3 : 1;                         // bf16 planes per operand
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16 *const As0 = reinterpret_cast<__bf16 *>(smem);
    __bf16 *const Bs0 = As0 + NP * BM * LDKB;
    constexpr int kBufStride = NP * (BM + BN) * LDKB;          // bf16 elements
    constexpr int kPlaneA = BM * LDKB, kPlaneB = BN * LDKB;
    int2 *const klut = reinterpret_cast<int2 *>(As0 + 2 * kBufStride);
    if constexpr (!FAST) {
        for (int kc = threadIdx.x; kc < a.k_pad / 4; kc += kThreads) {
            const int k = kc * 4;
            int2 e = make_int2(0, -1);
            if (k < a.K) {
                const int tap = k / a.cin, ci = k - tap * a.cin;
                const int kh = tap / a.kw, dy = kh * a.dil, dx = (tap - kh * a.kw) * a.dil;
                e = make_int2((dy * a.in_w + dx) * a.x_ld + ci, (dy << 16) | dx);
            }
            klut[kc] = e;
        }
        __syncthreads();
    }
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // Default walk: the output-channel tile changes slowest (consecutive workgroups share a weight tile).  korder bit 1
    // (desc.tile | SGV3D_TILE_MFIRST): the m-tile changes slowest -- its input rows are fetched from HBM once instead of once
    // per channel tile (tools/layer_traffic.py); pays on the bandwidth-heavy layers (64->256 at 216x384, 2560->512), costs on
    // the others, so it is one more candidate of the first-call measurement.
    int tn, tm;
    if (a.korder & 2) {
        tm = (int)((unsigned)logical / (unsigned)a.tiles_n);
        tn = logical - tm * a.tiles_n;
    } else {
        tn = (int)((unsigned)logical / (unsigned)a.tiles_m);     // (unsigned: half the scalar instructions of a signed division)
        tm = logical - tn * a.tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x;
    const int cc = tid & 7;
    const int r0 = tid >> 3;
    const int st_off = (((cc >> 1) ^ ((r0 >> 2) & 3)) * 8) + (cc & 1) * 4;   // swizzled place of this thread's 4 k in its row
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.w_bytes, 0x00020000);
    // The prologue runs in the shadow of the other workgroups' MFMAs: at four waves per SIMD an instruction of this wave issues
    // every 8-13 cycles, and ~1000 scalar + vector instructions of address set-up were 9-13 k cycles before the first k-tile
    // reached LDS (tools/igemm_stamps.py) -- as long as the whole k loop of a 256-deep 1x1 layer.  Hence the shortcuts below.
    const bool pointwise = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.in_h == a.m_h && a.in_w == a.m_w;
    unsigned a_off[A_CH];
    int a_ih0[A_CH], a_iw0[A_CH];
    bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        if (pointwise) {             // 1x1 / stride 1 / no padding: the input pixel IS the output pixel, no division
            a_ih0[i] = 0;
            a_iw0[i] = 0;
            a_off[i] = (unsigned)(((long long)mm * a.x_ld + a.x_coff + cc * 4) * XES);
            continue;
        }
        const int t = (int)((unsigned)mm / (unsigned)a.m_w);
        const int ow = mm - t * a.m_w;
        const int n = (int)((unsigned)t / (unsigned)a.m_h);
        const int oh = t - n * a.m_h;
        a_ih0[i] = oh * a.stride - a.pad;
        a_iw0[i] = ow * a.stride - a.pad;
        a_off[i] = (unsigned)((((long long)(n * a.in_h + a_ih0[i]) * a.in_w + a_iw0[i]) * a.x_ld + a.x_coff + cc * 4) * XES);
    }
    // WB: with bf16 activations the packed weights are the bf16 copy as well (half the L2 -> CU bytes of the operand every
    // workgroup of a column re-reads: 80 % of the requests of a ResNet 1x1 layer), stored to LDS as loaded
    constexpr bool WB = XB || YB;
    constexpr int WES = WB ? 2 : 4;
    unsigned b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) b_off[i] = (unsigned)(((size_t)(n0 + r0 + 32 * i) * a.k_pad + cc * 4) * WES);

    // bf16 MFMAs are 16x faster than f32 ones: the address arithmetic of the operand loads must shrink with them.
    // FAST path: which taps of a row's pixel fall inside the image is a bit mask computed once (<= 32 taps); a k-tile
    // then costs one bit test, one add and one select per 16-byte chunk.  1x1 / unpadded single-tap layers (most
    // launches of a ResNet) need nothing per tile: the lane offset is fixed and the channel chunk goes into the
    // buffer load's scalar offset.
    const int taps = a.kh * a.kw;
    const bool use_mask = FAST && taps <= 32;
    const bool one_tap = FAST && taps == 1 && a.pad == 0;
    unsigned a_mask[A_CH], a_fix[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        unsigned mk = 0;
        if (use_mask) {
            int t = 0;                               // (nested counters: no division per tap in the prologue)
            for (int th = 0; th < a.kh; ++th) {
                const int ih = a_ih0[i] + th * a.dil;
                const bool row_in = a_ok[i] && (unsigned)ih < (unsigned)a.in_h;
                for (int tw = 0; tw < a.kw; ++tw, ++t) {
                    const int iw = a_iw0[i] + tw * a.dil;
                    mk |= (row_in && (unsigned)iw < (unsigned)a.in_w) ? (1u << t) : 0u;
                }
            }
        }
        a_mask[i] = mk;
        a_fix[i] = (mk & 1u) ? a_off[i] : 0xffffffffu;
    }
    float4 ra0[A_CH], rb0[B_CH], ra1[A_CH], rb1[B_CH];
    const int nkt_all = FAST ? (a.cin / BK) * a.kh * a.kw : a.k_pad / BK;
    int kt_begin = 0, nkt = nkt_all;                                   // (no division without split-K: see the f32 kernel)
    if (a.split_k > 1) {
        kt_begin = (int)((unsigned)nkt_all * blockIdx.y / (unsigned)a.split_k);
        nkt = (int)((unsigned)nkt_all * (blockIdx.y + 1) / (unsigned)a.split_k);
    }
    int ld_kt = kt_begin, ld_kh = 0, ld_kw = 0, ld_c0 = 0;
    if constexpr (FAST) {
        if (kt_begin != 0) {
            const unsigned per_chunk = a.kh * a.kw;
            const unsigned kt0 = kt_begin < nkt_all ? kt_begin : nkt_all - 1;
            const unsigned chunk = kt0 / per_chunk, rem = kt0 - chunk * per_chunk;
            ld_c0 = chunk * BK;
            ld_kh = rem / (unsigned)a.kw;
            ld_kw = rem - ld_kh * a.kw;
        }
    }
    // one chunk of 4 k of the input: 16 bytes of f32, or (XB) 8 bytes of bf16 parked in .x / .y
#define SGV3D_LOAD_A(R, VOFF, SOFF)                                                                   \
    do {                                                                                              \
        if constexpr (XB) {                                                                           \
            const f32x2n t_ = __builtin_bit_cast(f32x2n, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, VOFF, SOFF, 0)); \
            R.x = t_.x; R.y = t_.y;                                                                   \
        } else {                                                                                      \
            const f32x4n t_ = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, VOFF, SOFF, 0)); \
            R.x = t_.x; R.y = t_.y; R.z = t_.z; R.w = t_.w;                                           \
        }                                                                                             \
    } while (0)
#define SGV3D_LOAD_TILE(RA, RB)                                                                       \
    do {                                                                                              \
        int dy_, dx_, koff_;                                                                          \
        bool kvalid_ = ld_kt < nkt;                                                                   \
        const int ktb_ = ld_kt < nkt ? ld_kt : nkt - 1;                                               \
        if constexpr (FAST) {                                                                         \
            dy_ = ld_kh * a.dil;                                                                      \
            dx_ = ld_kw * a.dil;                                                                      \
            koff_ = (dy_ * a.in_w + dx_) * a.x_ld + ld_c0;                                            \
        } else {                                                                                      \
            const int2 e_ = klut[ktb_ * (BK / 4) + cc];                                               \
            kvalid_ = kvalid_ & (e_.y >= 0);                                                          \
            dy_ = e_.y >> 16;                                                                         \
            dx_ = e_.y & 0xffff;                                                                      \
            koff_ = e_.x - cc * 4;                                                                    \
        }                                                                                             \
        if (one_tap) {                                                                                \
            _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                        \
                SGV3D_LOAD_A(RA[i], a_fix[i], koff_ * XES);                                           \
            }                                                                                         \
        } else if (use_mask) {                                                                        \
            const unsigned bit_ = kvalid_ ? 1u << (ld_kh * a.kw + ld_kw) : 0u;                        \
            _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                        \
                const unsigned vo_ = (a_mask[i] & bit_) ? a_off[i] + (unsigned)(koff_ * XES) : 0xffffffffu; \
                SGV3D_LOAD_A(RA[i], vo_, 0);                                                          \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                        \
                const int ih_ = a_ih0[i] + dy_, iw_ = a_iw0[i] + dx_;                                 \
                const bool v_ = a_ok[i] & kvalid_ & ((unsigned)ih_ < (unsigned)a.in_h) &              \
                                ((unsigned)iw_ < (unsigned)a.in_w);                                   \
                const unsigned vo_ = v_ ? a_off[i] + (unsigned)(koff_ * XES) : 0xffffffffu;           \
                SGV3D_LOAD_A(RA[i], vo_, 0);                                                          \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                                            \
            if constexpr (WB) {                                                                       \
                const f32x2n t_ = __builtin_bit_cast(f32x2n, __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, b_off[i], ktb_ * (BK * 2), 0)); \
                RB[i].x = t_.x; RB[i].y = t_.y;                                                       \
            } else {                                                                                  \
                const f32x4n t_ = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_off[i], ktb_ * (BK * 4), 0)); \
                RB[i].x = t_.x; RB[i].y = t_.y; RB[i].z = t_.z; RB[i].w = t_.w;                       \
            }                                                                                         \
        }                                                                                             \
        ++ld_kt;                                                                                      \
        if constexpr (FAST) {                                                                         \
            if (ld_kt < nkt) {                                                                        \
                if (++ld_kw == a.kw) {                                                                \
                    ld_kw = 0;                                                                        \
                    if (++ld_kh == a.kh) { ld_kh = 0; ld_c0 += BK; }                                  \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define SGV3D_CVT_STORE(DST, V, PLANE)                                                                \
    do {                                                                                              \
        f32x4v f_ = {V.x, V.y, V.z, V.w};                                                             \
        const bf16x4 h_ = __builtin_convertvector(f_, bf16x4);                                        \
        *reinterpret_cast<bf16x4 *>(DST) = h_;                                                        \
        if constexpr (SPLIT3) {                                                                       \
            f_ -= __builtin_convertvector(h_, f32x4v);                                                \
            const bf16x4 m_ = __builtin_convertvector(f_, bf16x4);                                    \
            *reinterpret_cast<bf16x4 *>((DST) + (PLANE)) = m_;                                        \
            f_ -= __builtin_convertvector(m_, f32x4v);                                                \
            *reinterpret_cast<bf16x4 *>((DST) + 2 * (PLANE)) = __builtin_convertvector(f_, bf16x4);   \
        }                                                                                             \
    } while (0)
#define SGV3D_STORE_TILE(RA, RB, BUF)                                                                 \
    do {                                                                                              \
        __bf16 *As_ = As0 + (BUF) * kBufStride, *Bs_ = Bs0 + (BUF) * kBufStride;                      \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                            \
            if constexpr (XB) {                                                                       \
                f32x2n t2_ = {RA[i].x, RA[i].y};                                                      \
                *reinterpret_cast<f32x2n *>(As_ + (r0 + 32 * i) * LDKB + st_off) = t2_;               \
            } else {                                                                                  \
                SGV3D_CVT_STORE(As_ + (r0 + 32 * i) * LDKB + st_off, RA[i], kPlaneA);                 \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                                            \
            if constexpr (WB) {                                                                       \
                f32x2n t2_ = {RB[i].x, RB[i].y};                                                      \
                *reinterpret_cast<f32x2n *>(Bs_ + (r0 + 32 * i) * LDKB + st_off) = t2_;               \
            } else {                                                                                  \
                SGV3D_CVT_STORE(Bs_ + (r0 + 32 * i) * LDKB + st_off, RB[i], kPlaneB);                 \
            }                                                                                         \
        }                                                                                             \
    } while (0)

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int a_frag_off = (wm * (BM / 2) + lr) * LDKB;
    const int b_frag_off = (wn * (BN / 2) + lr) * LDKB;
    const int rd_swz = (lr >> 2) & 3;                               // rows of a tile start at multiples of 32
    const int rd_off[2] = {((0 + lh) ^ rd_swz) * 8, ((2 + lh) ^ rd_swz) * 8};   // k-step s: logical 16-byte chunk 2 s + h
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    bf16x8 fa[NP][WTM], fb[NP][WTN];
#define SGV3D_READ_STEP(BUF, S)                                                                       \
    do {                                                                                              \
        _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) {                                           \
            _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt)                                        \
                fa[p_][mt] = *reinterpret_cast<const bf16x8 *>(As0 + (BUF) * kBufStride + p_ * kPlaneA + a_frag_off + mt * 32 * LDKB + rd_off[S]); \
            _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt)                                        \
                fb[p_][nt] = *reinterpret_cast<const bf16x8 *>(Bs0 + (BUF) * kBufStride + p_ * kPlaneB + b_frag_off + nt * 32 * LDKB + rd_off[S]); \
        }                                                                                             \
    } while (0)
#define SGV3D_MM(PA, PB)                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt)                                                \
        _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt)                                            \
            acc[mt][nt] = YB ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][nt], fa[PA][mt], acc[mt][nt], 0, 0, 0)     \
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][mt], fb[PB][nt], acc[mt][nt], 0, 0, 0);
    // small terms first, so that the leading product meets an accumulator that already holds the corrections
#define SGV3D_MFMA_STEP()                                                                             \
    do {                                                                                              \
        if constexpr (SPLIT3) {                                                                       \
            SGV3D_MM(0, 2) SGV3D_MM(2, 0) SGV3D_MM(1, 1) SGV3D_MM(0, 1) SGV3D_MM(1, 0)                \
        }                                                                                             \
        SGV3D_MM(0, 0)                                                                                \
    } while (0)
#define SGV3D_PHASE(BUF, RA, RB, SA, SB, HAVE_NEXT)                                                   \
    do {                                                                                              \
        SGV3D_LOAD_TILE(RA, RB);                                                                      \
        SGV3D_READ_STEP(BUF, 0);                                                                      \
        SGV3D_MFMA_STEP();                                                                            \
        SGV3D_READ_STEP(BUF, 1);                                                                      \
        if (HAVE_NEXT) SGV3D_STORE_TILE(SA, SB, (BUF) ^ 1);                                           \
        SGV3D_MFMA_STEP();                                                                            \
        __syncthreads();                                                                              \
    } while (0)

    // Residual prefetch (64x64 tile, fast-path epilogue): the small-K layers that carry a residual (the expanding 1x1
    // convolutions of the bottlenecks) spend a third of a workgroup's life waiting for the residual rows they only
    // ask for after the k loop; asked for here, the 16 values per lane arrive under the loop.
    constexpr bool kPrefetchRes = WTM * WTN <= 2 && !YB;
    // YB: lane -> (pixel p = lane >> 2 (+16 in the second pass), 8-channel chunk c = lane & 3) of a 32 x 32 tile
    constexpr bool kPrefetchResB = YB && WTM * WTN <= 2;
    const __bf16 *const resb = reinterpret_cast<const __bf16 *>(a.res);
    bf16x8 resq[WTM][WTN][2];
#define SGV3D_FETCH_RESB()                                                                            \
    do {                                                                                              \
        if (a.res != nullptr && a.split_k <= 1) {                                                     \
            _Pragma("unroll") for (int mt = 0; mt < WTM; ++mt)                                        \
                _Pragma("unroll") for (int nt = 0; nt < WTN; ++nt)                                    \
                    _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                \
                        const int row = m0 + (tid >> 7) * (BM / 2) + mt * 32 + ((tid & 63) >> 2) + 16 * ps; \
                        const int ch = n0 + ((tid >> 6) & 1) * (BN / 2) + nt * 32 + 8 * (tid & 3);    \
                        if (row < a.M && ch < a.N) resq[mt][nt][ps] = *reinterpret_cast<const bf16x8 *>(resb + (size_t)row * a.res_ld + ch); \
                    }                                                                                 \
        }                                                                                             \
    } while (0)
    // (GROUP_PLANES -- the per-branch hidden maps of the two-kernel head path -- is row-linear too when a wave's columns
    // stay inside one group: plane base + row * group width)
    const bool fast_epi = (a.split_k > 1 || (a.mode == SGV3D_CONV_NORMAL && a.gate == nullptr) ||
                           (a.mode == SGV3D_CONV_GROUP_PLANES && a.gate == nullptr && a.ks % (BN / 2) == 0)) && m0 + BM <= a.M;
    SGV3D_LOAD_TILE(ra0, rb0);
    SGV3D_LOAD_TILE(ra1, rb1);
    float resv[WTM][WTN][16];
    if constexpr (kPrefetchRes) {
        const bool want = fast_epi && a.split_k <= 1 && a.res != nullptr;
        const long long tile_row = m0 + __builtin_amdgcn_readfirstlane(wm) * (BM / 2);
        const __amdgpu_buffer_rsrc_t pr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(want ? a.res + tile_row * a.res_ld : a.zeros), 0, want ? (int)0xffffff00u : 0, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt) {
            const int col = n0 + wn * (BN / 2) + nt * 32 + lr;
            const unsigned roff0 = (want && col < a.N) ? (4u * lh * (unsigned)a.res_ld + col) * 4u : 0xffffffffu;
#pragma unroll
            for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned r_ = mt * 32 + (e & 3) + 8 * (e >> 2);
                    resv[mt][nt][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr_rsrc, roff0, r_ * a.res_ld * 4u, 0));
                }
        }
    }
    if constexpr (kPrefetchResB) SGV3D_FETCH_RESB();   // after the k-tiles: memory returns in order (see the f32 kernel)
    SGV3D_STORE_TILE(ra0, rb0, 0);
    __syncthreads();
    for (int kt = kt_begin; kt < nkt; kt += 2) {
        SGV3D_PHASE(0, ra0, rb0, ra1, rb1, kt + 1 < nkt);
        if (kt + 1 >= nkt) break;
        SGV3D_PHASE(1, ra1, rb1, ra0, rb0, kt + 2 < nkt);
    }
#undef SGV3D_LOAD_TILE
#undef SGV3D_LOAD_A
#undef SGV3D_CVT_STORE
#undef SGV3D_STORE_TILE
#undef SGV3D_PHASE
#undef SGV3D_READ_STEP
#undef SGV3D_MM
#undef SGV3D_MFMA_STEP

    if constexpr (YB) {
        // transposed accumulators: acc[mt][nt][4 g + i] = C[pixel m0 + wm (BM/2) + 32 mt + lr][channel n0 + wn (BN/2) + 32 nt + 8 g + 4 lh + i]
        const int prow0 = m0 + wm * (BM / 2), pcol0 = n0 + wn * (BN / 2);
        if (a.split_k > 1) {            // raw partial sums, [split][M][N] f32: 16-byte stores of 4 channels
            float *ws = a.ws + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
            for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
                for (int nt = 0; nt < WTN; ++nt) {
                    const int row = prow0 + mt * 32 + lr;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = pcol0 + nt * 32 + 8 * g + 4 * lh;
                        if (row < a.M && ch < a.N) {
                            f32x4n v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                            *reinterpret_cast<f32x4n *>(ws + (size_t)row * a.N + ch) = v;
                        }
                    }
                }
            return;
        }
        // 128x128 tile: no registers for the residual under the k loop, but the operand stages are dead now -- all residual
        // rows of the wave's four tiles are asked for at once, not one tile at a time behind the previous tile's stores
        if constexpr (YB && !kPrefetchResB) SGV3D_FETCH_RESB();
        constexpr int SLD = 36;                                          // floats per staged pixel row (144 B)
        float *stage = smem + wave * (32 * SLD);                          // the operand buffers are dead: the loop ended on a barrier
        __bf16 *const yb = reinterpret_cast<__bf16 *>(a.y);
        const int pc = lane & 3, pp = lane >> 2;
        const bool deconv = (a.mode & kConvModeMask) == SGV3D_CONV_DECONV;
        // per-channel terms of all the wave's n-tiles first: one memory latency, not one per n-tile
        int co_[WTN], dy_[WTN], dx_[WTN];
        f32x4n sc0_[WTN], sc1_[WTN], sh0_[WTN], sh1_[WTN];
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt) {
            const int ch = pcol0 + nt * 32 + 8 * pc;                      // GEMM column of this lane's 8-channel chunk
            // DECONV (kernel == stride transposed conv as a 1x1 GEMM): column = tap * cout + co, the chunk stays inside one
            // tap because cout % 8 == 0; input pixel (ih, iw) lands at output pixel (ih ks + dy, iw ks + dx)
            co_[nt] = ch; dy_[nt] = 0; dx_[nt] = 0;
            if (deconv) {
                const int tap = ch / a.cout;
                co_[nt] = ch - tap * a.cout;
                dy_[nt] = tap / a.ks;
                dx_[nt] = tap - dy_[nt] * a.ks;
            }
            const f32x4n one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
            sc0_[nt] = sc1_[nt] = one;
            sh0_[nt] = sh1_[nt] = zero;
            if (ch < a.N && a.scale) { sc0_[nt] = *reinterpret_cast<const f32x4n *>(a.scale + co_[nt]); sc1_[nt] = *reinterpret_cast<const f32x4n *>(a.scale + co_[nt] + 4); }
            if (ch < a.N && a.bias) { sh0_[nt] = *reinterpret_cast<const f32x4n *>(a.bias + co_[nt]); sh1_[nt] = *reinterpret_cast<const f32x4n *>(a.bias + co_[nt] + 4); }
        }
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt) {
            const int ch = pcol0 + nt * 32 + 8 * pc;
            const bool ch_ok = ch < a.N;
            const int co = co_[nt], dy = dy_[nt], dx = dx_[nt];
            const f32x4n sc0 = sc0_[nt], sc1 = sc1_[nt], sh0 = sh0_[nt], sh1 = sh1_[nt];
#pragma unroll
            for (int mt = 0; mt < WTM; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4n v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    *reinterpret_cast<f32x4n *>(stage + lr * SLD + 8 * g + 4 * lh) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int p = pp + 16 * ps;
                    const int row = prow0 + mt * 32 + p;
                    f32x4n v0 = *reinterpret_cast<const f32x4n *>(stage + p * SLD + 8 * pc);
                    f32x4n v1 = *reinterpret_cast<const f32x4n *>(stage + p * SLD + 8 * pc + 4);
                    if (row < a.M && ch_ok) {
                        v0 = v0 * sc0 + sh0;
                        v1 = v1 * sc1 + sh1;
                        if (a.res != nullptr) {
                            const bf16x8 rq = resq[mt][nt][ps];
#pragma unroll
                            for (int i = 0; i < 4; ++i) { v0[i] += (float)rq[i]; v1[i] += (float)rq[4 + i]; }
                        }
                        if (a.relu) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], 0.f); v1[i] = fmaxf(v1[i], 0.f); }
                        }
                        size_t yi = (size_t)row * a.y_ld;
                        if (deconv) {
                            const int t = row / a.m_w, iw = row - t * a.m_w;
                            const int img = t / a.m_h, ih = t - img * a.m_h;
                            yi = ((size_t)(img * a.out_h + ih * a.ks + dy) * a.out_w + (iw * a.ks + dx)) * a.y_ld;
                        }
                        const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
                        *reinterpret_cast<bf16x8 *>(yb + yi + a.y_coff + co) = __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
#undef SGV3D_FETCH_RESB
    // epilogue: the accumulator layout is that of every 32x32 MFMA (row = (e & 3) + 8 (e >> 2) + 4 h, column = lane & 31),
    // so the f32 kernel's fast path (hoisted channel terms, buffer stores with scalar row offsets) applies unchanged
    if (fast_epi) {
        const bool partial = a.split_k > 1;
        const int wmu = __builtin_amdgcn_readfirstlane(wm);
        const bool planes = !partial && a.mode == SGV3D_CONV_GROUP_PLANES;
        const int grp = planes ? (n0 + __builtin_amdgcn_readfirstlane(wn) * (BN / 2)) / a.ks : 0;
        const int col_sub = grp * a.ks;                      // columns are counted inside the group's plane
        const unsigned ld = partial ? (unsigned)a.N : planes ? (unsigned)a.ks : (unsigned)a.y_ld;
        const long long tile_row = m0 + wmu * (BM / 2);
        const float *const ybase = partial ? a.ws + ((size_t)blockIdx.y * a.M + tile_row) * a.N
                                   : planes ? a.y + ((size_t)grp * a.M + tile_row) * a.ks
                                            : a.y + tile_row * a.y_ld + a.y_coff;
        const __amdgpu_buffer_rsrc_t y_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)ybase, 0, (int)0xffffff00u, 0x00020000);
        const bool has_res = !partial && a.res != nullptr;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(has_res ? a.res + tile_row * a.res_ld : a.zeros), 0, has_res ? (int)0xffffff00u : 0, 0x00020000);
        const float floor_ = (!partial && a.relu) ? 0.f : -__builtin_inff();
        unsigned voff[WTN], roff[WTN];
        float sc[WTN], sh[WTN];
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt) {
            const int col = n0 + wn * (BN / 2) + nt * 32 + lr;
            const bool ok = col < a.N;
            voff[nt] = ok ? (4u * lh * ld + (col - col_sub)) * 4u : 0xffffffffu;
            roff[nt] = ok ? (4u * lh * (unsigned)a.res_ld + col) * 4u : 0xffffffffu;
            sc[nt] = (!partial && ok && a.scale) ? a.scale[col] : 1.f;
            sh[nt] = (!partial && ok && a.bias) ? a.bias[col] : 0.f;
        }
#define SGV3D_EPI_F(MT, NT, E)                                                                        \
    {                                                                                                 \
        const unsigned r_ = (MT) * 32 + ((E) & 3) + 8 * ((E) >> 2);                                   \
        float v_ = acc[MT][NT][E];                                                                    \
        if (!partial) {                                                                               \
            v_ = v_ * sc[NT] + sh[NT];                                                                \
            if constexpr (kPrefetchRes) {                                                             \
                v_ += resv[MT][NT][E]; /* zeros when there is no residual */                          \
            } else if (has_res)                                                                       \
                v_ += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, roff[NT], r_ * a.res_ld * 4u, 0)); \
            v_ = fmaxf(v_, floor_);                                                                   \
        }                                                                                             \
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), y_rsrc, voff[NT], r_ * ld * 4u, 0); \
    }
#define SGV3D_EPI_FT(MT, NT)                                                                          \
    SGV3D_EPI_F(MT, NT, 0) SGV3D_EPI_F(MT, NT, 1) SGV3D_EPI_F(MT, NT, 2) SGV3D_EPI_F(MT, NT, 3)       \
    SGV3D_EPI_F(MT, NT, 4) SGV3D_EPI_F(MT, NT, 5) SGV3D_EPI_F(MT, NT, 6) SGV3D_EPI_F(MT, NT, 7)       \
    SGV3D_EPI_F(MT, NT, 8) SGV3D_EPI_F(MT, NT, 9) SGV3D_EPI_F(MT, NT, 10) SGV3D_EPI_F(MT, NT, 11)     \
    SGV3D_EPI_F(MT, NT, 12) SGV3D_EPI_F(MT, NT, 13) SGV3D_EPI_F(MT, NT, 14) SGV3D_EPI_F(MT, NT, 15)
        SGV3D_EPI_FT(0, 0)
        if constexpr (WTN > 1) { SGV3D_EPI_FT(0, 1) }
        if constexpr (WTM > 1) {
            SGV3D_EPI_FT(1, 0)
            if constexpr (WTN > 1) { SGV3D_EPI_FT(1, 1) }
        }
#undef SGV3D_EPI_FT
#undef SGV3D_EPI_F
        return;
    }
    float *ws = a.split_k > 1 ? a.ws + (size_t)blockIdx.y * a.M * a.N : nullptr;
    const int row_base = m0 + wm * (BM / 2) + 4 * lh;
    const int col_base = n0 + wn * (BN / 2) + lr;
#define SGV3D_EPI_E(MT, NT, E)                                                                        \
    {                                                                                                 \
        const int col_ = col_base + (NT) * 32;                                                        \
        const int row_ = row_base + (MT) * 32 + ((E) & 3) + 8 * ((E) >> 2);                           \
        if (col_ < a.N && row_ < a.M) {                                                               \
            if (ws) ws[(size_t)row_ * a.N + col_] = acc[MT][NT][E];                                   \
            else conv_epilogue_store(a, row_, col_, acc[MT][NT][E]);                                  \
        }                                                                                             \
    }
#define SGV3D_EPI_TILE(MT, NT)                                                                        \
    SGV3D_EPI_E(MT, NT, 0) SGV3D_EPI_E(MT, NT, 1) SGV3D_EPI_E(MT, NT, 2) SGV3D_EPI_E(MT, NT, 3)       \
    SGV3D_EPI_E(MT, NT, 4) SGV3D_EPI_E(MT, NT, 5) SGV3D_EPI_E(MT, NT, 6) SGV3D_EPI_E(MT, NT, 7)       \
    SGV3D_EPI_E(MT, NT, 8) SGV3D_EPI_E(MT, NT, 9) SGV3D_EPI_E(MT, NT, 10) SGV3D_EPI_E(MT, NT, 11)     \
    SGV3D_EPI_E(MT, NT, 12) SGV3D_EPI_E(MT, NT, 13) SGV3D_EPI_E(MT, NT, 14) SGV3D_EPI_E(MT, NT, 15)
    SGV3D_EPI_TILE(0, 0)
    if constexpr (WTN > 1) { SGV3D_EPI_TILE(0, 1) }
    if constexpr (WTM > 1) {
        SGV3D_EPI_TILE(1, 0)
        if constexpr (WTN > 1) { SGV3D_EPI_TILE(1, 1) }
    }
#undef SGV3D_EPI_TILE
#undef SGV3D_EPI_E
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float *__restrict__ src, int cout, int cin, int kh, int kw, int cin_pad,
                                   int transposed, int korder, float *__restrict__ dst, int k_pad, int cout_pad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)k_pad * cout_pad) return;
    const int n = (int)(i / k_pad), k = (int)(i - (long long)n * k_pad);
    float v = 0.f;
    if (!transposed) {
        const int taps = kh * kw;
        const int K = taps * cin_pad;
        if (n < cout && k < K) {
            int tap, ci;
            if (korder == 1) {                       // channel-chunk-major: (chunk, tap, ci % 32)
                const int chunk = k / (taps * 32), rem = k - chunk * taps * 32;
                tap = rem >> 5;
                ci = chunk * 32 + (rem & 31);
            } else {
                tap = k / cin_pad;
                ci = k - tap * cin_pad;
            }
            const int y = tap / kw, x = tap - y * kw;
            if (ci < cin) v = src[(((size_t)n * cin + ci) * kh + y) * kw + x];
        }
    } else {
        // ConvTranspose2d weight [cin, cout, ks, ks]; GEMM row n = (dy*ks + dx)*cout + co, k = ci
        const int ks = kh;
        if (n < cout * ks * ks && k < cin_pad) {
            const int tap = n / cout, co = n - tap * cout;
            const int y = tap / ks, x = tap - y * ks;
            if (k < cin) v = src[(((size_t)k * cout + co) * ks + y) * ks + x];
        }
    }
    dst[i] = v;
}

// SGV3D_SWAP_EPI=1 selects the swapped-operand kernel where it applies (results are bitwise the same either way).  It is NOT the
// default: measured on the cfg-2 bottleneck layers (a round-4 probe, retired in round 6; us per launch alone / with three launches in
// flight) it ties or loses -- 64->256 @216x384 + residual 48.3 -> 55.1 alone, 128->512 @108x192 41.2 / 32.5 -> 43.2 / 34.7,
// 256->1024 @54x96 37.7 / 28.7 -> 38.8 / 29.0, 512->2048 @27x48 37.8 / 28.4 -> 35.1 / 27.4: a dword store instruction of the
// unswapped layout writes two full 128-byte row segments, a dwordx4 one of the swapped layout 32 bytes to each of 32 rows, and
// the memory pipeline prefers the former by as much as the four-fold drop in instruction count gains.
static bool swap_epilogue_enabled() {
    const char *e = getenv("SGV3D_SWAP_EPI");        // (read per launch: tests/test_conv_gpu.py flips it inside one process)
    return e && e[0] == '1';
}

// SGV3D_NO_PW_KERNEL=1: never the pointwise specialisation (A/B measurements; results are bitwise the same)
static bool pointwise_kernel_enabled() {
    const char *e = getenv("SGV3D_NO_PW_KERNEL");
    return !(e && e[0] == '1');
}

template <int WTM, int WTN, bool FAST>
int launch_t(const ConvArgs &a, hipStream_t st) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    constexpr size_t tiles_lds = sizeof(float) * 2 * (BM + BN) * LDK;
    // tap-major K: + the k-chunk decode table (8 bytes per 16-byte chunk of K)
    const size_t lds = tiles_lds + (FAST ? 0 : (size_t)a.k_pad / 4 * 8);
    SGV3D_REQUIRE(lds <= 160 * 1024, "conv2d_forward: K = %d too long for the tap-major kernel's decode table", a.K);
    ConvArgs b = a;
    b.zeros = conv_zero_block();
    if (!b.zeros) return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot resolve the zero block");
    b.tiles_m = cdiv(a.M, BM);
    b.tiles_n = cdiv(a.N, BN);
    if constexpr (WTM == 1 && WTN == 1 && FAST) {
        if (a.korder & 4) {      // five workgroups per CU (desc.tile | SGV3D_TILE_OCC5): 32 KB of LDS, one register stage
            constexpr size_t lds5 = sizeof(float) * 2 * (BM + BN) * BK;
            const bool pw5 = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.in_h == a.m_h && a.in_w == a.m_w &&
                             (a.mode & kConvModeMask) != SGV3D_CONV_DECONV;
            static PerDeviceSize lds_set_o, lds_set_opw;
            if (pw5) {
                if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<1, 1, true, false, true, false, true>), lds5, lds_set_opw))
                    return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot raise the dynamic LDS limit to %zu", lds5);
                hipLaunchKernelGGL((conv_igemm_kernel<1, 1, true, false, true, false, true>), dim3(b.tiles_m * b.tiles_n, b.split_k),
                                   dim3(kThreads), lds5, st, b);
            } else {
                if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<1, 1, true, false, false, false, true>), lds5, lds_set_o))
                    return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot raise the dynamic LDS limit to %zu", lds5);
                hipLaunchKernelGGL((conv_igemm_kernel<1, 1, true, false, false, false, true>), dim3(b.tiles_m * b.tiles_n, b.split_k),
                                   dim3(kThreads), lds5, st, b);
            }
            if (b.split_k > 1) return launch_splitk_reduce(b, st);
            return check_launch("conv_igemm_kernel(occ5)");
        }
        // 16-byte epilogue (operands swapped): NHWC output in 16-byte channel quads, everything 16-B aligned
        const bool quads = a.N % 4 == 0 && a.y_ld % 4 == 0 && a.y_coff % 4 == 0 && (a.res == nullptr || a.res_ld % 4 == 0) &&
                           ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res) |
                             reinterpret_cast<uintptr_t>(a.scale) | reinterpret_cast<uintptr_t>(a.bias)) & 15) == 0;
        if (quads && a.split_k <= 1 && a.mode == SGV3D_CONV_NORMAL && a.gate == nullptr && swap_epilogue_enabled()) {
            static PerDeviceSize lds_set_swap;
            if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<1, 1, true, true>), lds, lds_set_swap))
                return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot raise the dynamic LDS limit to %zu", lds);
            hipLaunchKernelGGL((conv_igemm_kernel<1, 1, true, true>), dim3(b.tiles_m * b.tiles_n, 1), dim3(kThreads), lds, st, b);
            return check_launch("conv_igemm_kernel(swap)");
        }
    }
    if constexpr (WTM == 1 && FAST) {
        // pointwise specialisation (64x64 and 64x128 tiles): no vector instruction for the A addresses inside the loop
        const bool pw = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.in_h == a.m_h && a.in_w == a.m_w &&
                        (a.mode & kConvModeMask) != SGV3D_CONV_DECONV && a.cin >= 128 && pointwise_kernel_enabled();
        // (cin >= 128: with two k-tiles -- the 64 -> 256 expanders at 216x384, HBM-bound -- the generic kernel's longer
        //  prologue spreads the residual and output traffic better: 47.8 vs 53.5 us alone, 42.6 vs 43.7 with three in flight;
        //  everywhere else the specialisation wins 3-6 %: the same probe)
        if (pw) {
            static PerDeviceSize lds_set_pw;
            if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<WTM, WTN, true, false, true>), lds, lds_set_pw))
                return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot raise the dynamic LDS limit to %zu", lds);
            hipLaunchKernelGGL((conv_igemm_kernel<WTM, WTN, true, false, true>), dim3(b.tiles_m * b.tiles_n, b.split_k),
                               dim3(kThreads), lds, st, b);
            if (b.split_k > 1) return launch_splitk_reduce(b, st);
            return check_launch("conv_igemm_kernel(pointwise)");
        }
    }
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<WTM, WTN, FAST>), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv2d_forward: cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL((conv_igemm_kernel<WTM, WTN, FAST>), dim3(b.tiles_m * b.tiles_n, b.split_k), dim3(kThreads), lds,
                       st, b);
    if (b.split_k > 1) return launch_splitk_reduce(b, st);
    return check_launch("conv_igemm_kernel");
}

// 32 x 128 tile, pointwise (conv_gemm_grouped only)
int launch_narrow_pw(const ConvArgs &a, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * 2 * (32 + 128) * LDK;
    ConvArgs b = a;
    b.zeros = conv_zero_block();
    if (!b.zeros) return fail(SGV3D_ELAUNCH, "conv_gemm_grouped: cannot resolve the zero block");
    b.tiles_m = cdiv(a.M, 32);
    b.tiles_n = cdiv(a.N, 128);
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel<1, 1, true, false, true, true>), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv_gemm_grouped: cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL((conv_igemm_kernel<1, 1, true, false, true, true>), dim3(b.tiles_m * b.tiles_n, 1), dim3(kThreads), lds, st, b);
    return check_launch("conv_igemm_kernel(narrow)");
}

template <int WTM, int WTN>
int launch(const ConvArgs &a, hipStream_t st) {
    return (a.korder & 1) ? launch_t<WTM, WTN, true>(a, st) : launch_t<WTM, WTN, false>(a, st);
}

template <int WTM, int WTN, bool FAST, bool SPLIT3, bool XB = false, bool YB = false>
int launch_bf16_t(const ConvArgs &a, hipStream_t st) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    constexpr size_t tiles_only = sizeof(unsigned short) * 2 * (SPLIT3 ? 3 : 1) * (BM + BN) * LDKB;
    constexpr size_t stage_lds = YB ? sizeof(float) * 4 * 32 * 36 : 0;       // YB epilogue: a 32 x 36 f32 tile per wave
    constexpr size_t tiles_lds = tiles_only > stage_lds ? tiles_only : stage_lds;
    const size_t lds = tiles_lds + (FAST ? 0 : (size_t)a.k_pad / 4 * 8);
    SGV3D_REQUIRE(lds <= 160 * 1024, "conv2d_forward_bf16: K = %d too long for the tap-major kernel's decode table", a.K);
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_bf16_kernel<WTM, WTN, FAST, SPLIT3, XB, YB>), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv2d_forward_bf16: cannot raise the dynamic LDS limit to %zu", lds);
    ConvArgs b = a;
    b.zeros = conv_zero_block();
    if (!b.zeros) return fail(SGV3D_ELAUNCH, "conv2d_forward_bf16: cannot resolve the zero block");
    b.tiles_m = cdiv(a.M, BM);
    b.tiles_n = cdiv(a.N, BN);
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<WTM, WTN, FAST, SPLIT3, XB, YB>), dim3(b.tiles_m * b.tiles_n, b.split_k), dim3(kThreads),
                       lds, st, b);
    if (b.split_k > 1) return launch_splitk_reduce(b, st);
    return check_launch("conv_igemm_bf16_kernel");
}

// io: bit 0 = the input is bf16, bit 1 = output and residual are bf16
template <int WTM, int WTN>
int launch_bf16io(const ConvArgs &a, hipStream_t st, int io) {
    const bool fast = (a.korder & 1) != 0;
    switch (io) {
        case 1: return fast ? launch_bf16_t<WTM, WTN, true, false, true, false>(a, st) : launch_bf16_t<WTM, WTN, false, false, true, false>(a, st);
        case 2: return fast ? launch_bf16_t<WTM, WTN, true, false, false, true>(a, st) : launch_bf16_t<WTM, WTN, false, false, false, true>(a, st);
        case 3: return fast ? launch_bf16_t<WTM, WTN, true, false, true, true>(a, st) : launch_bf16_t<WTM, WTN, false, false, true, true>(a, st);
        default: return fail(SGV3D_EINVAL, "conv2d_forward_bf16io: io flags %d", io);
    }
}

template <int WTM, int WTN>
int launch_bf16(const ConvArgs &a, hipStream_t st, bool split3) {
    if (split3) return (a.korder & 1) ? launch_bf16_t<WTM, WTN, true, true>(a, st) : launch_bf16_t<WTM, WTN, false, true>(a, st);
    return (a.korder & 1) ? launch_bf16_t<WTM, WTN, true, false>(a, st) : launch_bf16_t<WTM, WTN, false, false>(a, st);
}

int pick_tile(long long M, int N) {
    // cost = (max workgroups any CU runs) x tile area x a small-tile inefficiency factor
    const int bm[4] = {128, 128, 64, 64}, bn[4] = {128, 64, 128, 64};
    const double pen[4] = {1.0, 1.06, 1.06, 1.18};
    int best = 0;
    double best_cost = 1e300;
    for (int t = 0; t < 4; ++t) {
        const long long tiles = ((M + bm[t] - 1) / bm[t]) * ((N + bn[t] - 1) / bn[t]);
        const double rounds = (double)((tiles + 255) / 256);
        const double cost = rounds * bm[t] * bn[t] * pen[t];
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best + 1;
}

}  // namespace

namespace sgv3d {
int launch_splitk_reduce(const ConvArgs &a, hipStream_t st) {
    const long long total = (long long)a.M * a.N;
    const bool quads = a.mode == SGV3D_CONV_NORMAL && a.gate == nullptr && a.N % 4 == 0 && a.y_ld % 4 == 0 && a.y_coff % 4 == 0 &&
                       (a.res == nullptr || a.res_ld % 4 == 0) && total / 4 < 0x7fffffffLL &&
                       ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res) | reinterpret_cast<uintptr_t>(a.ws) |
                         reinterpret_cast<uintptr_t>(a.scale) | reinterpret_cast<uintptr_t>(a.bias)) & 15) == 0;
    if (quads) {
        hipLaunchKernelGGL(conv_splitk_reduce4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, a);
        return check_launch("conv_splitk_reduce4_kernel");
    }
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, a);
    return check_launch("conv_splitk_reduce_kernel");
}
const float *conv_zero_block() {
    static const float *zero_block[kMaxDevices] = {};      // a device symbol has one address per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    const float *z = __atomic_load_n(&zero_block[dev], __ATOMIC_RELAXED);
    if (!z) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero16)) != hipSuccess) return nullptr;
        z = static_cast<const float *>(p);
        __atomic_store_n(&zero_block[dev], z, __ATOMIC_RELAXED);
    }
    return z;
}
}  // namespace sgv3d

extern "C" void sgv3d_conv_pack_geometry(int k, int n, int *k_pad, int *n_pad) {
    if (k_pad) *k_pad = ((k + BK - 1) / BK) * BK;
    if (n_pad) *n_pad = ((n + 127) / 128) * 128;
}

extern "C" int sgv3d_conv_pack_weight(const float *w_src, int cout, int cin, int kh, int kw, int cin_pad,
                                      int transposed, int k_order, float *w_packed, int k_pad, int cout_pad,
                                      void *stream) {
    SGV3D_REQUIRE(w_src && w_packed, "conv_pack_weight: null pointer");
    SGV3D_REQUIRE(cout > 0 && cin > 0 && kh > 0 && kw > 0 && cin_pad >= cin, "conv_pack_weight: bad shape");
    const int K = transposed ? cin_pad : kh * kw * cin_pad;
    const int Nn = transposed ? cout * kh * kw : cout;
    SGV3D_REQUIRE(!transposed || kh == kw, "conv_pack_weight: transposed needs a square kernel");
    SGV3D_REQUIRE(k_order == 0 || (k_order == 1 && cin_pad % BK == 0), "conv_pack_weight: k_order 1 needs cin %% 32 == 0");
    SGV3D_REQUIRE(k_pad >= K && k_pad % BK == 0 && cout_pad >= Nn && cout_pad % 128 == 0,
                  "conv_pack_weight: k_pad=%d / cout_pad=%d do not cover K=%d, N=%d", k_pad, cout_pad, K, Nn);
    const long long total = (long long)k_pad * cout_pad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, cout, cin,
                       kh, kw, cin_pad, transposed, k_order, w_packed, k_pad, cout_pad);
    return check_launch("pack_weight_kernel");
}

namespace {
__global__ __launch_bounds__(256) void weight_to_bf16_kernel(const float *__restrict__ src, long long n, __bf16 *__restrict__ dst) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (__bf16)src[i];
}
}  // namespace

extern "C" int sgv3d_conv_weight_to_bf16(const float *w_packed, int k_pad, int cout_pad, void *w_packed_bf16, void *stream) {
    SGV3D_REQUIRE(w_packed && w_packed_bf16 && k_pad > 0 && cout_pad > 0, "conv_weight_to_bf16: bad arguments");
    const long long n = (long long)k_pad * cout_pad;
    hipLaunchKernelGGL(weight_to_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w_packed, n,
                       static_cast<__bf16 *>(w_packed_bf16));
    return check_launch("weight_to_bf16_kernel");
}

#ifdef SGV3D_IGEMM_STAMPS
extern "C" int sgv3d_igemm_debug_stamps(void *buf) {
    long long *p = static_cast<long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

namespace sgv3d {
int conv_gemm_grouped(const float *x, const float *w, float *y, int rows, int groups, int K, int N, int k_pad, int cout_pad,
                      int k_order, int tile, hipStream_t st) {
    SGV3D_REQUIRE(x && w && y && rows > 0 && groups > 0 && K > 0 && N > 0, "conv_gemm_grouped: bad argument");
    const bool occ5 = (tile & SGV3D_TILE_OCC5) != 0;      // (with the 64x64 tile: its five-workgroups-per-CU form)
    tile &= ~SGV3D_TILE_OCC5;
    SGV3D_REQUIRE(!occ5 || (tile == SGV3D_TILE_64x64 && k_order == 1), "conv_gemm_grouped: SGV3D_TILE_OCC5 goes with the 64x64 tile and k_order 1");
    const bool narrow = tile == SGV3D_TILE_32x128;
    SGV3D_REQUIRE(rows % (narrow ? 32 : 64) == 0 && K % 4 == 0 && k_pad >= K && k_pad % BK == 0 && cout_pad >= N && cout_pad % 128 == 0,
                  "conv_gemm_grouped: rows %% 64 (32 for the narrow tile), K %% 4, k_pad / cout_pad as packed (rows=%d K=%d k_pad=%d N=%d cout_pad=%d)",
                  rows, K, k_pad, N, cout_pad);
    SGV3D_REQUIRE(!narrow || (k_order == 1 && K % BK == 0 && K >= 128), "conv_gemm_grouped: the narrow tile needs K %% 32 == 0, K >= 128");
    SGV3D_REQUIRE(k_order == 0 || K % BK == 0, "conv_gemm_grouped: k_order 1 needs K %% 32 == 0");
    const long long M = (long long)rows * groups;
    SGV3D_REQUIRE(M < 0x7fffffffLL && M * K * 4 < 0xf0000000LL && (long long)cout_pad * k_pad * 4 < 0xf0000000LL,
                  "conv_gemm_grouped: operands larger than 3.75 GiB (32-bit buffer offsets)");
    ConvArgs a;
    a.zeros = nullptr;
    a.x = x; a.w = w; a.scale = nullptr; a.bias = nullptr; a.res = nullptr; a.gate = nullptr; a.y = y;
    a.in_h = 1; a.in_w = (int)M; a.cin = K; a.out_h = 1; a.out_w = (int)M; a.cout = N;
    a.m_h = 1; a.m_w = (int)M;
    a.kh = a.kw = 1; a.stride = 1; a.pad = 0; a.dil = 1;
    a.x_ld = K; a.x_coff = 0; a.y_ld = N; a.y_coff = 0; a.res_ld = 0; a.relu = 0; a.mode = SGV3D_CONV_NORMAL; a.ks = 0;
    a.k_pad = k_pad; a.tiles_m = a.tiles_n = 0;
    a.korder = k_order | (occ5 ? 4 : 0);
    a.x_bytes = (unsigned)(M * K * 4);
    a.w_bytes = (unsigned)((long long)cout_pad * k_pad * 4);          // one group's block
    a.M = (int)M; a.N = N; a.K = K;
    a.split_k = 1; a.ws = nullptr;
    a.wb_y = rows; a.wb_x = cout_pad * k_pad;
    switch (tile) {
        case SGV3D_TILE_32x128: return launch_narrow_pw(a, st);
        case SGV3D_TILE_64x128: return launch<1, 2>(a, st);
        case SGV3D_TILE_64x64: return launch<1, 1>(a, st);
        default: return fail(SGV3D_EINVAL, "conv_gemm_grouped: tile must be 64x64, 64x128 or 32x128 (got %d)", tile);
    }
}
}  // namespace sgv3d

extern "C" size_t sgv3d_conv2d_workspace_bytes(const sgv3d_conv_desc *d) {
    if (!d || d->split_k <= 1) return 0;
    const long long mh = d->mode == SGV3D_CONV_DECONV ? d->in_h : d->out_h;
    const long long mw = d->mode == SGV3D_CONV_DECONV ? d->in_w : d->out_w;
    const long long n = d->mode == SGV3D_CONV_DECONV ? (long long)d->cout * d->deconv_ks * d->deconv_ks : d->cout;
    return sizeof(float) * (size_t)d->split_k * d->batch * mh * mw * n;
}

static int conv2d_forward_impl(const sgv3d_conv_desc *d, const float *x, const float *w_packed,
                               const float *scale, const float *bias, const float *residual,
                               const float *gate, float *y, void *workspace, size_t workspace_bytes,
                               void *stream, int bf16 /* 0: f32 MFMA, 1: bf16 operands, 3: f32 as three bf16 terms */,
                               int io = 0 /* bf16 == 1 only: bit 0 = x is a bf16 tensor, bit 1 = y and residual are */) {
    SGV3D_REQUIRE(d && x && w_packed && y, "conv2d_forward: null pointer");
    SGV3D_REQUIRE(io == 0 || bf16 == 1, "conv2d_forward: bf16 tensors need the bf16 MFMA entry point");
    if (io & 2) {
        SGV3D_REQUIRE((d->mode == SGV3D_CONV_NORMAL || d->mode == SGV3D_CONV_DECONV) && gate == nullptr,
                      "conv2d_forward_bf16io: bf16 output in NORMAL / DECONV mode without gate only");
        SGV3D_REQUIRE(d->cout % 8 == 0 && d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && (residual == nullptr || d->res_ld % 8 == 0),
                      "conv2d_forward_bf16io: cout / y_ld / y_coff / res_ld must be multiples of 8 (16-byte rows of bf16)");
        SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0,
                      "conv2d_forward_bf16io: y, residual, scale and bias must be 16-B aligned");
    }
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->out_h > 0 && d->out_w > 0 &&
                      d->cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0,
                  "conv2d_forward: non-positive dimension");
    SGV3D_REQUIRE((d->cin & 3) == 0 && (d->x_ld & 3) == 0 && (d->x_coff & 3) == 0,
                  "conv2d_forward: cin/x_ld/x_coff must be multiples of 4 (got %d/%d/%d)", d->cin, d->x_ld, d->x_coff);
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_packed) & 15) == 0,
                  "conv2d_forward: x and w_packed must be 16-B aligned");
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin, "conv2d_forward: x_ld too small");
    SGV3D_REQUIRE(d->x_nchw == 0, "conv2d_forward: NCHW input is ingested with sgv3d_nchw_to_nhwc first");
    SGV3D_REQUIRE(residual == nullptr || d->res_ld >= d->cout, "conv2d_forward: res_ld too small");
    ConvArgs a;
    a.zeros = nullptr;
    a.wb_y = a.wb_x = 0;        // (not a grouped GEMM)
    a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.gate = gate; a.y = y;
    a.in_h = d->in_h; a.in_w = d->in_w; a.cin = d->cin; a.out_h = d->out_h; a.out_w = d->out_w; a.cout = d->cout;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld;
    a.relu = d->relu; a.mode = d->mode; a.ks = d->deconv_ks; a.k_pad = d->k_pad;
    a.tiles_m = a.tiles_n = 0;
    a.korder = d->k_order | ((d->tile & SGV3D_TILE_MFIRST) ? 2 : 0) | ((d->tile & SGV3D_TILE_OCC5) ? 4 : 0);
    SGV3D_REQUIRE(!(d->tile & SGV3D_TILE_OCC5) || ((d->tile & ~(SGV3D_TILE_MFIRST | SGV3D_TILE_OCC5)) == SGV3D_TILE_64x64 &&
                                                    (d->k_order & 1) && !io && !bf16),
                  "conv2d_forward: SGV3D_TILE_OCC5 goes with the f32 64x64 tile and channel-chunk-major weights (cin %% 32 == 0)");
    {
        const long long xb = (long long)d->batch * d->in_h * d->in_w * d->x_ld * ((io & 1) ? 2 : 4), wb = (long long)d->cout_pad * d->k_pad * (io ? 2 : 4);
        SGV3D_REQUIRE(xb < 0xf0000000LL && wb < 0xf0000000LL, "conv2d_forward: input / packed weights larger than 3.75 GiB (32-bit buffer offsets)");
        a.x_bytes = (unsigned)xb;
        a.w_bytes = (unsigned)wb;
    }
    SGV3D_REQUIRE(d->k_order == 0 || (d->k_order == 1 && d->cin % BK == 0), "conv2d_forward: k_order 1 needs cin %% 32 == 0");
    if (d->mode == SGV3D_CONV_DECONV) {
        SGV3D_REQUIRE(d->deconv_ks >= 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0,
                      "conv2d_forward: DECONV runs as a 1x1 GEMM with deconv_ks = kernel = stride");
        SGV3D_REQUIRE(d->out_h == d->in_h * d->deconv_ks && d->out_w == d->in_w * d->deconv_ks,
                      "conv2d_forward: DECONV output must be input * deconv_ks");
        SGV3D_REQUIRE(residual == nullptr && gate == nullptr, "conv2d_forward: DECONV has no residual/gate");
        a.m_h = d->in_h; a.m_w = d->in_w;
        a.N = d->cout * d->deconv_ks * d->deconv_ks;
    } else {
        SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL || d->mode == SGV3D_CONV_NCHW_OUT || d->mode == SGV3D_CONV_GROUP_PLANES,
                      "conv2d_forward: bad mode %d", d->mode);
        SGV3D_REQUIRE(d->mode != SGV3D_CONV_GROUP_PLANES || (d->deconv_ks > 0 && d->cout % d->deconv_ks == 0 && !residual),
                      "conv2d_forward: GROUP_PLANES needs deconv_ks = group width dividing cout, no residual");
        const int eh = (d->in_h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int ew = (d->in_w + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        SGV3D_REQUIRE(eh == d->out_h && ew == d->out_w, "conv2d_forward: output %dx%d does not match conv arithmetic %dx%d",
                      d->out_h, d->out_w, eh, ew);
        SGV3D_REQUIRE(d->mode != SGV3D_CONV_NCHW_OUT || (residual == nullptr), "conv2d_forward: NCHW_OUT has no residual");
        a.m_h = d->out_h; a.m_w = d->out_w;
        a.N = d->cout;
    }
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NCHW_OUT || d->mode == SGV3D_CONV_GROUP_PLANES || d->y_ld >= d->y_coff + d->cout,
                  "conv2d_forward: y_ld too small");
    const long long M = (long long)d->batch * a.m_h * a.m_w;
    SGV3D_REQUIRE(M < 0x7fffffffLL, "conv2d_forward: too many output pixels");
    SGV3D_REQUIRE((long long)d->batch * d->in_h * d->in_w * d->x_ld < (1LL << 40), "conv2d_forward: input too large");
    a.M = (int)M;
    a.K = d->kh * d->kw * d->cin;
    SGV3D_REQUIRE(d->k_pad >= a.K && d->k_pad % BK == 0, "conv2d_forward: k_pad=%d does not cover K=%d", d->k_pad, a.K);
    SGV3D_REQUIRE(d->cout_pad >= a.N && d->cout_pad % 128 == 0, "conv2d_forward: cout_pad=%d does not cover N=%d",
                  d->cout_pad, a.N);
    a.split_k = d->split_k > 1 ? d->split_k : 1;
    a.ws = static_cast<float *>(workspace);
    SGV3D_REQUIRE(a.split_k <= d->k_pad / BK && a.split_k <= 64, "conv2d_forward: split_k=%d too large for %d k-tiles",
                  a.split_k, d->k_pad / BK);
    if (a.split_k > 1) {
        const size_t need = sizeof(float) * (size_t)a.split_k * a.M * a.N;
        if (!workspace || workspace_bytes < need)
            return fail(SGV3D_ENOSPACE, "conv2d_forward: split-K workspace has %zu bytes, needs %zu", workspace_bytes, need);
    }
    const int tile_bits = d->tile & ~(SGV3D_TILE_MFIRST | SGV3D_TILE_OCC5);
    const int tile = tile_bits ? tile_bits : pick_tile(M, a.N);
    hipStream_t st = as_stream(stream);
    if (io & 2) a.mode |= kConvYBf16 | kConvResBf16;       // (NORMAL / DECONV mode, checked above)
    if (io) {
        switch (tile) {
            case SGV3D_TILE_128x128: return launch_bf16io<2, 2>(a, st, io);
            case SGV3D_TILE_128x64: return launch_bf16io<2, 1>(a, st, io);
            case SGV3D_TILE_64x128: return launch_bf16io<1, 2>(a, st, io);
            case SGV3D_TILE_64x64: return launch_bf16io<1, 1>(a, st, io);
            default: return fail(SGV3D_EINVAL, "conv2d_forward_bf16io: unknown tile %d", tile);
        }
    }
    if (bf16) {
        switch (tile) {
            case SGV3D_TILE_128x128: return launch_bf16<2, 2>(a, st, bf16 == 3);
            case SGV3D_TILE_128x64: return launch_bf16<2, 1>(a, st, bf16 == 3);
            case SGV3D_TILE_64x128: return launch_bf16<1, 2>(a, st, bf16 == 3);
            case SGV3D_TILE_64x64: return launch_bf16<1, 1>(a, st, bf16 == 3);
            default: return fail(SGV3D_EINVAL, "conv2d_forward_bf16: unknown tile %d", tile);
        }
    }
    switch (tile) {
        case SGV3D_TILE_128x128: return launch<2, 2>(a, st);
        case SGV3D_TILE_128x64: return launch<2, 1>(a, st);
        case SGV3D_TILE_64x128: return launch<1, 2>(a, st);
        case SGV3D_TILE_64x64: return launch<1, 1>(a, st);
        default: return fail(SGV3D_EINVAL, "conv2d_forward: unknown tile %d", tile);
    }
}

extern "C" int sgv3d_conv2d_forward(const sgv3d_conv_desc *d, const float *x, const float *w_packed,
                                    const float *scale, const float *bias, const float *residual,
                                    const float *gate, float *y, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    return conv2d_forward_impl(d, x, w_packed, scale, bias, residual, gate, y, workspace, workspace_bytes, stream, 0);
}

extern "C" int sgv3d_conv2d_forward_bf16(const sgv3d_conv_desc *d, const float *x, const float *w_packed,
                                         const float *scale, const float *bias, const float *residual,
                                         const float *gate, float *y, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    return conv2d_forward_impl(d, x, w_packed, scale, bias, residual, gate, y, workspace, workspace_bytes, stream, 1);
}

extern "C" int sgv3d_conv2d_forward_bf16io(const sgv3d_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                           const float *bias, const void *residual, const float *gate, void *y,
                                           void *workspace, size_t workspace_bytes, void *stream, int io_flags) {
    SGV3D_REQUIRE(io_flags >= 1 && io_flags <= 3, "conv2d_forward_bf16io: io_flags must be 1 (bf16 input), 2 (bf16 output + residual) or 3");
    return conv2d_forward_impl(d, static_cast<const float *>(x), static_cast<const float *>(w_packed), scale, bias,
                               static_cast<const float *>(residual), gate, static_cast<float *>(y), workspace, workspace_bytes, stream,
                               1, io_flags);
}

extern "C" int sgv3d_conv2d_forward_f32x3(const sgv3d_conv_desc *d, const float *x, const float *w_packed,
                                          const float *scale, const float *bias, const float *residual,
                                          const float *gate, float *y, void *workspace, size_t workspace_bytes,
                                          void *stream) {
    return conv2d_forward_impl(d, x, w_packed, scale, bias, residual, gate, y, workspace, workspace_bytes, stream, 3);
}
