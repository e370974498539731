// Fused CenterHead branches on the bf16 matrix cores (gfx950) -- the bf16-mode counterpart of conv_wino_head_kernel.
//
// Replaces, for every one of the nb = 6 tasks x 6 branches of mmdet3d's SeparateHead (reference call site
// layers/heads/bev_height_head.py:110):   [3x3 conv 64 -> 64, BN, ReLU]  then  [3x3 conv 64 -> c, bias]
// (c <= 4).  The two-kernel bf16 path wrote the 36 hidden maps to HBM (2.4 GB per cfg-3 frame) and read them back;
// here a workgroup owns a 16 x 16 pixel tile of the map, keeps its 20 x 20 x 64 input patch in LDS as bf16 for its
// whole lifetime and walks over the branches:
//   layer 1  implicit GEMM  M = 18 x 18 hidden pixels (the tile + the one-pixel ring layer 2 needs; recomputing the
//            ring costs 27 % more MFMA work and no HBM traffic, no fix-up pass), N = 64, K = 9 taps x 64 channels on
//            v_mfma_f32_32x32x16_bf16.  A fragments: ds_read_b128 from the (XOR-swizzled) patch image; B fragments:
//            the weights are packed on the host in fragment order, so a lane's fragment is one 16-byte global load
//            straight into registers (all workgroups stream the same 72 KB per branch: L2-resident), prefetched one
//            tap ahead.  No barrier inside the tap loop.
//   epilogue folded BN + ReLU in fp32, zero outside the image (layer 2 pads the hidden map with zeros), rounded to bf16
//            into the hidden image in LDS.
//   layer 2  M = 256 pixels (one 16-pixel row per MFMA tile), N = 16 (c padded), K = 576 on v_mfma_f32_16x16x32_bf16;
//            bias added in fp32, 16-byte stores into the NCHW output planes.
// MFMA time per workgroup and branch: 216 + 72 instructions per wave; LDS traffic 5 ds_read_b128 per 6 MFMAs.
// Bound: MFMA bf16 (2.5 PFLOP/s dense); algorithmic work 2 * H*W * (nb*64*576 + total_out*576) flop per frame.
#include "common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kTP = 16;              // output tile (pixels per side)
constexpr int kHP = kTP + 2;         // hidden tile
constexpr int kIP = kTP + 4;         // input patch
constexpr int kCH = 64;              // channels of the shared map == hidden channels
// LDS images are [row][8 channel chunks][x][8 channels] (bf16): the 16 lanes of a ds_read_b128 service group are
// consecutive pixels of one row and one chunk = 256 contiguous bytes, conflict-free without any swizzle, and the tap /
// k-step part of every address is a compile-time constant that goes into the instruction's offset field.
constexpr int kInRowB = 8 * kIP * 16;            // bytes of one patch row (all chunks): 2560
constexpr int kInChunkB = kIP * 16;              // 320
constexpr int kInBytes = kIP * kInRowB;          // 51 200
constexpr int kHidRows = 22;                     // 18 real rows + 4 rows that absorb the 60 padding pixels of the 12 m-tiles
constexpr int kHidRowB = 8 * kHP * 16;           // 2304
constexpr int kHidChunkB = kHP * 16;             // 288
constexpr int kHidBytes = kHidRows * kHidRowB;   // 50 688
constexpr int kW2Elems = 9 * 4 * kCH;            // [tap][col < 4][64 ch] bf16
constexpr int kMaxBranches = 48;     // folded-BN table of all branches in LDS: 2 x 48 x 64 floats = 24 KB
constexpr int kHeadLds = kInBytes + kHidBytes + kW2Elems * 2 + 2 * kMaxBranches * kCH * 4;
constexpr int kMT = 3;               // 32-pixel m-tiles per wave in layer 1 (4 waves x 3 x 32 = 384 >= 324)
constexpr int kW1PerBranch = 9 * 4 * 2 * 64 * 8; // bf16 elements: [tap][k-step][n-tile][lane][8]

__global__ __launch_bounds__(64) void head_bf16_pack_w1_kernel(const float *__restrict__ w1 /* [nb*64][64][3][3] */, int nb,
                                                                __bf16 *__restrict__ out) {
    // one wave per (branch, tap, k-step, n-tile): lane l writes its 8-element fragment
    const int id = blockIdx.x;
    const int nt = id & 1, ks = (id >> 1) & 3, tap = (id >> 3) % 9, br = id / 72;
    if (br >= nb) return;
    const int l = threadIdx.x;
    const int cout = br * 64 + nt * 32 + (l & 31);
    __bf16 *dst = out + (size_t)id * 512 + l * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cin = ks * 16 + 8 * (l >> 5) + j;
        dst[j] = (__bf16)w1[((size_t)cout * 64 + cin) * 9 + tap];
    }
}

// final layers: f32 [total_out][3][3][64] -> bf16 [branch][tap][col < 4][64], zero rows for col >= c (one 4608-byte image
// per branch, copied to LDS as it is)
__global__ __launch_bounds__(256) void head_bf16_pack_w2_kernel(const float *__restrict__ w2, const int32_t *__restrict__ out_begin,
                                                                 int nb, __bf16 *__restrict__ out) {
    const int br = blockIdx.x;
    if (br >= nb) return;
    const int ob = out_begin[br], c = out_begin[br + 1] - ob;
    for (int e = threadIdx.x; e < kW2Elems; e += 256) {
        const int ch = e & 63, col = (e >> 6) & 3, tap = e >> 8;
        out[(size_t)br * kW2Elems + e] = col < c ? (__bf16)w2[((size_t)(ob + col) * 9 + tap) * 64 + ch] : (__bf16)0.f;
    }
}

struct HeadBf16Args {
    const float *x;            // NHWC fp32 [B, H, W, x_ld], channels [x_coff, x_coff + 64)
    const __bf16 *w1;          // packed fragments
    const float *scale1, *shift1;   // [nb * 64] folded BN of the first layers
    const __bf16 *w2;          // packed [nb][9][4][64] bf16
    const float *bias2;        // [total_out]
    const int32_t *out_begin;  // [nb + 1]
    float *out;                // NCHW [B, total_out, H, W]
    int H, W, x_ld, x_coff, nb, total_out, tiles_x, tiles_y;
    long long *dbg;            // optional [4 * nb + 2] timestamps of workgroup 0 (tools only), else NULL
    int x_bf16;                // 1: x is a bf16 tensor (bf16-activation mode)
};

__global__ __launch_bounds__(256, 1) void head_bf16_kernel(const HeadBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_s = smem;
    char *hid_s = smem + kInBytes;
    __bf16 *w2_s = reinterpret_cast<__bf16 *>(smem + kInBytes + kHidBytes);
    float *bn_s = reinterpret_cast<float *>(smem + kInBytes + kHidBytes + kW2Elems * 2);     // [2][kMaxBranches * 64]: scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int y0 = ty * kTP, x0 = tx * kTP;

    // ---- input patch: 400 pixels x 8 chunks of 8 channels (fp32 -> bf16); zeros outside the image ---------------------
    {
        const size_t xoff0 = (size_t)b * a.H * a.W * a.x_ld + a.x_coff;
        const float *xb = a.x + xoff0;
        for (int e = tid; e < kIP * kIP * 8; e += 256) {        // e = pixel * 8 + chunk: a pixel's 256 B are read by 8 lanes
            const int p = e >> 3, chunk = e & 7;
            const int iy = p / kIP, ix = p - iy * kIP;
            const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t eoff = ((size_t)gy * a.W + gx) * a.x_ld + chunk * 8;
                if (a.x_bf16) {
                    v = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(a.x) + xoff0 + eoff);
                } else {
                    const float *src = xb + eoff;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(src), hi = *reinterpret_cast<const f32x4 *>(src + 4);
                    const bf16x4 l4 = __builtin_convertvector(lo, bf16x4), h4 = __builtin_convertvector(hi, bf16x4);
                    v = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            *reinterpret_cast<bf16x8 *>(in_s + iy * kInRowB + chunk * kInChunkB + ix * 16) = v;
        }
        for (int e = tid; e < a.nb * kCH; e += 256) {
            bn_s[e] = a.scale1[e];
            bn_s[kMaxBranches * kCH + e] = a.shift1[e];
        }
    }
    // ---- per-lane geometry, fixed for the whole kernel ---------------------------------------------------------------------
    // layer 1 is computed transposed (C^T = W1 . X^T: the weight fragment is the A operand, the pixel fragment the B
    // operand), so an accumulator tile has the PIXEL on the lane (col = lane & 31) and 4 consecutive hidden channels in
    // registers 4g .. 4g+3 (channel = 8g + 4h + i).
    const char *ain[kMT];              // patch address of hidden pixel m at tap (0, 0), k-step 0, this lane's k half
    char *hout[kMT];                   // hidden-image address of pixel m, chunk 0, this lane's channel half
    bool hin[kMT];                     // hidden pixel m lies inside the image (else it is the zero padding of layer 2)
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
        const int m = wave * (kMT * 32) + mt * 32 + r;          // m >= 324: padding pixels, computed from a valid patch address
        const int hy = m / kHP, hx = m - hy * kHP;              // and parked in hidden rows 18..21, which nobody reads
        const int hyc = hy < kHP ? hy : kHP - 1;
        ain[mt] = in_s + hyc * kInRowB + h * kInChunkB + hx * 16;
        hout[mt] = hid_s + hy * kHidRowB + hx * 16 + h * 8;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        hin[mt] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    // layer 2: lane = (pixel x = lane & 15, k quarter kq = lane >> 4) for the A operand, (column = lane & 15, kq) for B
    const int col = lane & 15, kq = lane >> 4;
    const char *a2base = hid_s + (wave * 4) * kHidRowB + kq * kHidChunkB + col * 16;
    const __bf16 *b2base = w2_s + (col & 3) * 64 + kq * 8;     // columns >= 4 read a copy of columns 0..3: never stored
    const bf16x8 *w1l = reinterpret_cast<const bf16x8 *>(a.w1) + lane;   // + ((br*9 + tap)*4 + ks)*2 + nt) * 64
    __syncthreads();

    bf16x8 bq[2][4][2];                // weight fragments of the current / next tap: [buffer][k-step][n-tile]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = w1l[(size_t)ks * 2 * 64 + nt * 64];

#define HEAD_STAMP(slot) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.dbg[slot] = (long long)__builtin_readcyclecounter(); } while (0)
    HEAD_STAMP(0);
    for (int br = 0; br < a.nb; ++br) {
        f32x16 acc[kMT][2];
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        // this branch's final-layer weights (4608 B, bf16, packed on the host side): requested now, parked in LDS by the
        // epilogue -- 288 threads-worth of 16 bytes
        bf16x8 w2pre[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            if (e < kW2Elems / 8) w2pre[u] = reinterpret_cast<const bf16x8 *>(a.w2 + (size_t)br * kW2Elems)[e];
        }
        // ---------------------------------------------------------------- layer 1: 9 taps x 4 k-steps of 16 channels
        // Software pipeline pinned with scheduling barriers (left alone, the compiler sinks every weight-fragment load to
        // its first use and the kernel runs at the latency of a global load per MFMA pair): the 8 weight fragments of tap
        // t+1 are requested before the 24 MFMAs of tap t, the 3 pixel fragments of k-step s+1 before the 6 MFMAs of step s.
        bf16x8 af[2][kMT];
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) af[0][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1, nxt = cur ^ 1;
            {
                int nb_ = br, nt_ = tap + 1;
                if (nt_ == 9) { nt_ = 0; nb_ = br + 1 < a.nb ? br + 1 : br; }
                const size_t base = (size_t)((nb_ * 9 + nt_) * 4) * 2 * 64;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq[nxt][ks][nt] = w1l[base + (size_t)ks * 2 * 64 + nt * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ac = (tap * 4 + ks) & 1, an = ac ^ 1;
                if (!(tap == 8 && ks == 3)) {                      // pixel fragments of the next k-step (of the next tap after ks == 3)
                    const int t2 = ks == 3 ? tap + 1 : tap, k2 = ks == 3 ? 0 : ks + 1;
                    const int off2 = (t2 / 3) * kInRowB + (t2 % 3) * 16 + k2 * 2 * kInChunkB;     // immediate
#pragma unroll
                    for (int mt = 0; mt < kMT; ++mt) af[an][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt] + off2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[cur][ks][nt], af[ac][mt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // 9 taps: the last tap used buffer 0 and refilled buffer 1 with the next branch's tap 0 -> move it to buffer 0
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = bq[1][ks][nt];

        HEAD_STAMP(1 + br * 4);
        __syncthreads();               // every wave is done reading the previous branch's hidden image / w2
        HEAD_STAMP(2 + br * 4);
        // ---------------------------------------------------------------- epilogue 1: BN + ReLU -> hidden image (bf16)
        // straight-line: every lane stores (padding pixels go to rows nobody reads), the image mask is applied to the
        // packed result
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = nt * 32 + 8 * g + 4 * h;                       // 4 consecutive hidden channels
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(bn_s + br * 64 + ch);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(bn_s + kMaxBranches * kCH + br * 64 + ch);
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(acc[mt][nt][4 * g + i] * sc[i] + sh[i], 0.f);
                    union { bf16x4 b; unsigned long long u; } pk;
                    pk.b = __builtin_convertvector(v, bf16x4);
                    pk.u = hin[mt] ? pk.u : 0ull;
                    *reinterpret_cast<unsigned long long *>(hout[mt] + (nt * 4 + g) * kHidChunkB) = pk.u;
                }
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            if (e < kW2Elems / 8) reinterpret_cast<bf16x8 *>(w2_s)[e] = w2pre[u];
        }
        __syncthreads();
        HEAD_STAMP(3 + br * 4);
        // ---------------------------------------------------------------- layer 2: 4 pixel rows per wave, N = 16, K = 576
        {
            const int ob = a.out_begin[br], c = a.out_begin[br + 1] - ob;
            f32x4 acc2[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[rr][i] = 0.f;
            // all 18 weight fragments first, then the pixel fragments three steps ahead of their MFMAs (a step is 4 MFMAs =
            // 64 cycles, an LDS read takes ~130); every address is base + immediate
            bf16x8 bf[18];
#pragma unroll
            for (int st = 0; st < 18; ++st) bf[st] = *reinterpret_cast<const bf16x8 *>(b2base + (st >> 1) * 256 + (st & 1) * 32);
            constexpr int kAhead = 3;
            bf16x8 a2[kAhead + 1][4];
#define HEAD_A2(st_, rr_) *reinterpret_cast<const bf16x8 *>(a2base + ((rr_) + ((st_) >> 1) / 3) * kHidRowB + (((st_) >> 1) % 3) * 16 + ((st_) & 1) * 4 * kHidChunkB)
#pragma unroll
            for (int st = 0; st < kAhead; ++st)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) a2[st][rr] = HEAD_A2(st, rr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + kAhead < 18) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) a2[(st + kAhead) % (kAhead + 1)][rr] = HEAD_A2(st + kAhead, rr);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    acc2[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[st % (kAhead + 1)][rr], bf[st], acc2[rr], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // C: col = lane & 15 (output channel), row = (lane >> 4) * 4 + reg (pixel x inside the tile row)
            if (col < c) {
                const float bias = a.bias2[ob + col];
                float *plane = a.out + ((size_t)b * a.total_out + ob + col) * a.H * a.W;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int gy = y0 + wave * 4 + rr, gx = x0 + kq * 4;
                    if (gy < a.H) {
                        float *dst = plane + (size_t)gy * a.W + gx;
                        if (gx + 3 < a.W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                            f32x4 v = acc2[rr];
                            v += bias;
                            *reinterpret_cast<f32x4 *>(dst) = v;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (gx + i < a.W) dst[i] = acc2[rr][i] + bias;
                        }
                    }
                }
            }
        }
        HEAD_STAMP(4 + br * 4);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Warp-specialised variant (the default): 8 waves.  Waves 0..3 run layer 1 + its epilogue of branch i while waves 4..7 run
// layer 2 + the stores of branch i-1 from the other half of a double-buffered hidden image, and stage the folded-BN
// vectors / final-layer weights of the coming branches.  One workgroup barrier per branch.  The layer-2 phase (LDS
// bandwidth bound, 4.0 k cycles) and the loads hide behind layer 1 + epilogue (MFMA / VALU bound, 11.3 k cycles).
// Same arithmetic in the same order as head_bf16_kernel: results are bitwise identical.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kWsHidRows = kHP + 1;                               // 18 real rows + 1 row all padding pixels are parked in
constexpr int kWsHidBytes = kWsHidRows * kHidRowB;                // 43 776
constexpr int kWsLds = kInBytes + 2 * kWsHidBytes + 2 * kW2Elems * 2 + 2 * 2 * kCH * 4;   // 51 200 + 87 552 + 9 216 + 1 024

__global__ __launch_bounds__(512, 1) void head_bf16_ws_kernel(const HeadBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_s = smem;
    char *hid_s = smem + kInBytes;                                                     // [2][kWsHidBytes]
    __bf16 *w2_s = reinterpret_cast<__bf16 *>(smem + kInBytes + 2 * kWsHidBytes);      // [2][kW2Elems]
    float *bn_s = reinterpret_cast<float *>(smem + kInBytes + 2 * kWsHidBytes + 2 * kW2Elems * 2);   // [2][scale 64 | shift 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool l1 = wave < 4;                                       // wave-uniform role
    const int gw = wave & 3, gt = tid & 255;                        // wave / thread index inside the role group
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int y0 = ty * kTP, x0 = tx * kTP;
    {
        const size_t xoff0 = (size_t)b * a.H * a.W * a.x_ld + a.x_coff;
        const float *xb = a.x + xoff0;
        for (int e = tid; e < kIP * kIP * 8; e += 512) {
            const int p = e >> 3, chunk = e & 7;
            const int iy = p / kIP, ix = p - iy * kIP;
            const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t eoff = ((size_t)gy * a.W + gx) * a.x_ld + chunk * 8;
                if (a.x_bf16) {
                    v = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(a.x) + xoff0 + eoff);
                } else {
                    const float *src = xb + eoff;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(src), hi = *reinterpret_cast<const f32x4 *>(src + 4);
                    const bf16x4 l4 = __builtin_convertvector(lo, bf16x4), h4 = __builtin_convertvector(hi, bf16x4);
                    v = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            *reinterpret_cast<bf16x8 *>(in_s + iy * kInRowB + chunk * kInChunkB + ix * 16) = v;
        }
        if (tid < 2 * kCH) bn_s[tid] = tid < kCH ? a.scale1[tid] : a.shift1[tid - kCH];           // branch 0
    }
    // layer-1 geometry (waves 0..3; computed by everybody, cheap)
    const char *ain[kMT];
    int hoff[kMT];                     // byte offset inside one hidden buffer
    bool hin[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
        const int m = gw * (kMT * 32) + mt * 32 + r;
        const int hy = m / kHP, hx = m - hy * kHP;
        const int hyc = hy < kHP ? hy : kHP - 1;
        ain[mt] = in_s + hyc * kInRowB + h * kInChunkB + hx * 16;
        hoff[mt] = (hy < kHP ? hy : kHP) * kHidRowB + hx * 16 + h * 8;        // padding pixels -> row 18 (never read)
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        hin[mt] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    // the 324 hidden pixels are 10.125 m-tiles of 32 rows on 12: the last m-tile of the last layer-1 wave (rows 352 .. 383) is all
    // padding -- its MFMAs are skipped (the chip is at its power limit in this loop: MFMAs on padding cost the others clock)
    const bool last_tile_pad = gw * (kMT * 32) + (kMT - 1) * 32 >= kHP * kHP;
    const int col = lane & 15, kq = lane >> 4;
    const int a2off = (gw * 4) * kHidRowB + kq * kHidChunkB + col * 16;
    const int b2off = (col & 3) * 64 + kq * 8;
    const bf16x8 *w1l = reinterpret_cast<const bf16x8 *>(a.w1) + lane;
    bf16x8 bq[2][4][2];
    if (l1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = w1l[(size_t)ks * 2 * 64 + nt * 64];
    }
    __syncthreads();

    for (int it = 0; it <= a.nb; ++it) {
        if (l1) {
            if (it < a.nb) {
                const int br = it;
                char *hbuf = hid_s + (br & 1) * kWsHidBytes;
                const float *bnb = bn_s + (br & 1) * 2 * kCH;
                f32x16 acc[kMT][2];
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
                bf16x8 af[2][kMT];
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) af[0][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt]);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cur = tap & 1, nxt = cur ^ 1;
                    {
                        int nb_ = br, nt_ = tap + 1;
                        if (nt_ == 9) { nt_ = 0; nb_ = br + 1 < a.nb ? br + 1 : br; }
                        const size_t base = (size_t)((nb_ * 9 + nt_) * 4) * 2 * 64;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) bq[nxt][ks][nt] = w1l[base + (size_t)ks * 2 * 64 + nt * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int ac = (tap * 4 + ks) & 1, an = ac ^ 1;
                        if (!(tap == 8 && ks == 3)) {
                            const int t2 = ks == 3 ? tap + 1 : tap, k2 = ks == 3 ? 0 : ks + 1;
                            const int off2 = (t2 / 3) * kInRowB + (t2 % 3) * 16 + k2 * 2 * kInChunkB;
#pragma unroll
                            for (int mt = 0; mt < kMT; ++mt) af[an][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt] + off2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < kMT; ++mt) {
                            if (mt == kMT - 1 && last_tile_pad) continue;        // (wave-uniform)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[cur][ks][nt], af[ac][mt], acc[mt][nt], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = bq[1][ks][nt];
                // epilogue: BN + ReLU -> hidden buffer (br & 1); its previous reader (layer 2 of branch br - 2) finished
                // before the barrier that ended iteration br - 1
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = nt * 32 + 8 * g + 4 * h;
                        const f32x4 sc = *reinterpret_cast<const f32x4 *>(bnb + ch);
                        const f32x4 sh = *reinterpret_cast<const f32x4 *>(bnb + kCH + ch);
#pragma unroll
                        for (int mt = 0; mt < kMT; ++mt) {
                            f32x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(acc[mt][nt][4 * g + i] * sc[i] + sh[i], 0.f);
                            union { bf16x4 b; unsigned long long u; } pk;
                            pk.b = __builtin_convertvector(v, bf16x4);
                            pk.u = hin[mt] ? pk.u : 0ull;
                            *reinterpret_cast<unsigned long long *>(hbuf + hoff[mt] + (nt * 4 + g) * kHidChunkB) = pk.u;
                        }
                    }
            }
        } else {
            // ---- staging for later iterations: folded BN of branch it + 1 (read by the epilogue of iteration it + 1) and the
            // final-layer weights of branch it (read by layer 2 in iteration it + 1)
            if (it + 1 < a.nb && gt < 2 * kCH)
                bn_s[((it + 1) & 1) * 2 * kCH + gt] = gt < kCH ? a.scale1[(it + 1) * kCH + gt] : a.shift1[(it + 1) * kCH + gt - kCH];
            bf16x8 w2pre[2];
            if (it < a.nb) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = gt + u * 256;
                    if (e < kW2Elems / 8) w2pre[u] = reinterpret_cast<const bf16x8 *>(a.w2 + (size_t)it * kW2Elems)[e];
                }
            }
            if (it >= 1) {
                const int br = it - 1;
                const char *hbuf = hid_s + (br & 1) * kWsHidBytes;
                const __bf16 *w2b = w2_s + (br & 1) * kW2Elems;
                const int ob = a.out_begin[br], c = a.out_begin[br + 1] - ob;
                f32x4 acc2[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc2[rr][i] = 0.f;
                bf16x8 bf[18];
#pragma unroll
                for (int st = 0; st < 18; ++st) bf[st] = *reinterpret_cast<const bf16x8 *>(w2b + b2off + (st >> 1) * 256 + (st & 1) * 32);
                constexpr int kAhead = 3;
                bf16x8 a2[kAhead + 1][4];
#define HEAD_A2W(st_, rr_) *reinterpret_cast<const bf16x8 *>(hbuf + a2off + ((rr_) + ((st_) >> 1) / 3) * kHidRowB + (((st_) >> 1) % 3) * 16 + ((st_) & 1) * 4 * kHidChunkB)
#pragma unroll
                for (int st = 0; st < kAhead; ++st)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) a2[st][rr] = HEAD_A2W(st, rr);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + kAhead < 18) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) a2[(st + kAhead) % (kAhead + 1)][rr] = HEAD_A2W(st + kAhead, rr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        acc2[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[st % (kAhead + 1)][rr], bf[st], acc2[rr], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef HEAD_A2W
                if (col < c) {
                    const float bias = a.bias2[ob + col];
                    float *plane = a.out + ((size_t)b * a.total_out + ob + col) * a.H * a.W;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int gy = y0 + gw * 4 + rr, gx = x0 + kq * 4;
                        if (gy < a.H) {
                            float *dst = plane + (size_t)gy * a.W + gx;
                            if (gx + 3 < a.W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                                f32x4 v = acc2[rr];
                                v += bias;
                                *reinterpret_cast<f32x4 *>(dst) = v;
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (gx + i < a.W) dst[i] = acc2[rr][i] + bias;
                            }
                        }
                    }
                }
            }
            if (it < a.nb) {                      // w2 of branch `it` -> buffer it & 1 (its previous content, branch it - 2, was consumed in iteration it - 1)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = gt + u * 256;
                    if (e < kW2Elems / 8) reinterpret_cast<bf16x8 *>(w2_s + (it & 1) * kW2Elems)[e] = w2pre[u];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Ping-pong variant (built, measured, NOT the default: sgv3d_centerhead_bf16_select_plain(3)): 8 waves in two groups of four; group g owns the branches b = g (mod 2) and alternates between
// "layer 1 of my branch" and "epilogue + layer 2 + stores of my branch", half a period out of phase with the other group.  In the
// warp-specialised kernel above the layer-1 waves also run the epilogue (3.0 k of their 11.3 k cycles per branch: vector ALU and LDS
// stores, the matrix pipe idle) while the layer-2 waves are idle two thirds of the time; here every wave does both kinds of work
// and at any time one group is on the matrix pipe while the other is on the vector ALU / LDS: a half-step is
// max(4 taps, epilogue) + max(5 taps, layer 2) = 8.3 k cycles per branch.  Each group has its own hidden image, final-layer weights
// and folded-BN vectors in LDS (the double buffers of the kernel above); two workgroup barriers per half-step.  Same arithmetic in
// the same order as head_bf16_kernel: bitwise identical results.  MEASURED: 243 us for 36 branches at 256 x 256 -- exactly the
// warp-specialised kernel's 243 us (single-role 296): with all 256 CUs in a dense bf16 MFMA loop the chip is at its power / clock
// limit, and putting the epilogue under the other group's MFMAs buys no time.  What would: fewer MFMAs (the 27 % ring recompute and
// the 324 -> 384 row padding).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void head_bf16_pp_kernel(const HeadBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_s = smem;
    char *hid_s = smem + kInBytes;                                                     // [2][kWsHidBytes]
    __bf16 *w2_s = reinterpret_cast<__bf16 *>(smem + kInBytes + 2 * kWsHidBytes);      // [2][kW2Elems]
    float *bn_s = reinterpret_cast<float *>(smem + kInBytes + 2 * kWsHidBytes + 2 * kW2Elems * 2);   // [2][scale 64 | shift 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);      // two groups of four waves: even / odd branches
    const int gw = wave & 3, gt = tid & 255;                        // wave / thread index inside the role group
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int y0 = ty * kTP, x0 = tx * kTP;
    {
        const size_t xoff0 = (size_t)b * a.H * a.W * a.x_ld + a.x_coff;
        const float *xb = a.x + xoff0;
        for (int e = tid; e < kIP * kIP * 8; e += 512) {
            const int p = e >> 3, chunk = e & 7;
            const int iy = p / kIP, ix = p - iy * kIP;
            const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t eoff = ((size_t)gy * a.W + gx) * a.x_ld + chunk * 8;
                if (a.x_bf16) {
                    v = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(a.x) + xoff0 + eoff);
                } else {
                    const float *src = xb + eoff;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(src), hi = *reinterpret_cast<const f32x4 *>(src + 4);
                    const bf16x4 l4 = __builtin_convertvector(lo, bf16x4), h4 = __builtin_convertvector(hi, bf16x4);
                    v = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            *reinterpret_cast<bf16x8 *>(in_s + iy * kInRowB + chunk * kInChunkB + ix * 16) = v;
        }
        if (gt < 2 * kCH && grp < a.nb)                                                                 // folded BN of branch `grp`
            bn_s[grp * 2 * kCH + gt] = gt < kCH ? a.scale1[grp * kCH + gt] : a.shift1[grp * kCH + gt - kCH];
    }
    // layer-1 geometry (waves 0..3; computed by everybody, cheap)
    const char *ain[kMT];
    int hoff[kMT];                     // byte offset inside one hidden buffer
    bool hin[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
        const int m = gw * (kMT * 32) + mt * 32 + r;
        const int hy = m / kHP, hx = m - hy * kHP;
        const int hyc = hy < kHP ? hy : kHP - 1;
        ain[mt] = in_s + hyc * kInRowB + h * kInChunkB + hx * 16;
        hoff[mt] = (hy < kHP ? hy : kHP) * kHidRowB + hx * 16 + h * 8;        // padding pixels -> row 18 (never read)
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        hin[mt] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    const int col = lane & 15, kq = lane >> 4;
    const int a2off = (gw * 4) * kHidRowB + kq * kHidChunkB + col * 16;
    const int b2off = (col & 3) * 64 + kq * 8;
    const bf16x8 *w1l = reinterpret_cast<const bf16x8 *>(a.w1) + lane;
    char *const hbuf = hid_s + grp * kWsHidBytes;                 // the group's own hidden image
    float *const bnb = bn_s + grp * 2 * kCH;
    __bf16 *const w2b = w2_s + grp * kW2Elems;
    __syncthreads();

    // Group g owns the branches b = g, g + 2, ... and runs, per branch: layer 1 (a barrier in its middle, one at its end), then
    // epilogue -> barrier -> layer 2 + stores -> barrier.  Group 1 starts half a period (two barriers) late, so between any two
    // barriers one group is on the matrix pipe and the other on the vector ALU / LDS; trailing barriers equalise the counts.
    if (grp == 1) { __syncthreads(); __syncthreads(); }
    for (int br = grp; br < a.nb; br += 2) {
        f32x16 acc[kMT][2];
        {
            // (the fragments of tap 0 are requested here, not at the end of the group's previous layer 1: 32 registers that
            // would be live across the other role -- with them the kernel spills)
            bf16x8 bq[2][4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = w1l[(size_t)(br * 9 * 4) * 2 * 64 + (size_t)ks * 2 * 64 + nt * 64];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
            bf16x8 af[2][kMT];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) af[0][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap == 4) __syncthreads();               // the half-step's middle barrier (the other group: epilogue done)
                const int cur = tap & 1, nxt = cur ^ 1;
                if (tap < 8) {
                    const size_t base = (size_t)((br * 9 + tap + 1) * 4) * 2 * 64;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) bq[nxt][ks][nt] = w1l[base + (size_t)ks * 2 * 64 + nt * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ac = (tap * 4 + ks) & 1, an = ac ^ 1;
                    if (!(tap == 8 && ks == 3)) {
                        const int t2 = ks == 3 ? tap + 1 : tap, k2 = ks == 3 ? 0 : ks + 1;
                        const int off2 = (t2 / 3) * kInRowB + (t2 % 3) * 16 + k2 * 2 * kInChunkB;
#pragma unroll
                        for (int mt = 0; mt < kMT; ++mt) af[an][mt] = *reinterpret_cast<const bf16x8 *>(ain[mt] + off2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[cur][ks][nt], af[ac][mt], acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
        bf16x8 w2pre[2];
        float bnpre = 0.f;
        {
            // requested now, parked in LDS after the middle barrier: the final-layer weights of this branch and the folded BN of
            // the group's next branch (br + 2)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = gt + u * 256;
                if (e < kW2Elems / 8) w2pre[u] = reinterpret_cast<const bf16x8 *>(a.w2 + (size_t)br * kW2Elems)[e];
            }
            if (br + 2 < a.nb && gt < 2 * kCH) bnpre = gt < kCH ? a.scale1[(br + 2) * kCH + gt] : a.shift1[(br + 2) * kCH + gt - kCH];
            // epilogue: BN + ReLU -> the group's hidden image (its previous reader, layer 2 of branch br - 2, finished two
            // barriers ago)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = nt * 32 + 8 * g + 4 * h;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(bnb + ch);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(bnb + kCH + ch);
#pragma unroll
                    for (int mt = 0; mt < kMT; ++mt) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(acc[mt][nt][4 * g + i] * sc[i] + sh[i], 0.f);
                        union { bf16x4 b; unsigned long long u; } pk;
                        pk.b = __builtin_convertvector(v, bf16x4);
                        pk.u = hin[mt] ? pk.u : 0ull;
                        *reinterpret_cast<unsigned long long *>(hbuf + hoff[mt] + (nt * 4 + g) * kHidChunkB) = pk.u;
                    }
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = gt + u * 256;
                if (e < kW2Elems / 8) reinterpret_cast<bf16x8 *>(w2b)[e] = w2pre[u];
            }
        }
        __syncthreads();                                          // hidden image and final-layer weights complete
        {
            if (br + 2 < a.nb && gt < 2 * kCH) bnb[gt] = bnpre;        // (the epilogue of br has read its own)
            const int ob = a.out_begin[br], c = a.out_begin[br + 1] - ob;
            f32x4 acc2[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[rr][i] = 0.f;
            bf16x8 bf[18];
#pragma unroll
            for (int st = 0; st < 18; ++st) bf[st] = *reinterpret_cast<const bf16x8 *>(w2b + b2off + (st >> 1) * 256 + (st & 1) * 32);
            constexpr int kAhead = 3;
            bf16x8 a2[kAhead + 1][4];
#define HEAD_A2W(st_, rr_) *reinterpret_cast<const bf16x8 *>(hbuf + a2off + ((rr_) + ((st_) >> 1) / 3) * kHidRowB + (((st_) >> 1) % 3) * 16 + ((st_) & 1) * 4 * kHidChunkB)
#pragma unroll
            for (int st = 0; st < kAhead; ++st)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) a2[st][rr] = HEAD_A2W(st, rr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + kAhead < 18) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) a2[(st + kAhead) % (kAhead + 1)][rr] = HEAD_A2W(st + kAhead, rr);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    acc2[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[st % (kAhead + 1)][rr], bf[st], acc2[rr], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef HEAD_A2W
            if (col < c) {
                const float bias = a.bias2[ob + col];
                float *plane = a.out + ((size_t)b * a.total_out + ob + col) * a.H * a.W;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int gy = y0 + gw * 4 + rr, gx = x0 + kq * 4;
                    if (gy < a.H) {
                        float *dst = plane + (size_t)gy * a.W + gx;
                        if (gx + 3 < a.W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                            f32x4 v = acc2[rr];
                            v += bias;
                            *reinterpret_cast<f32x4 *>(dst) = v;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (gx + i < a.W) dst[i] = acc2[rr][i] + bias;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (grp == (a.nb & 1)) { __syncthreads(); __syncthreads(); }
}

}  // namespace

static long long *g_head_dbg = nullptr;
static int g_head_plain = 0;       // 0: warp-specialised kernel (default), 1: single-role kernel, 3: ping-pong kernel
// tools / tests only: 1 selects the single-role kernel (4 waves, the phases of a branch run one after the other), 3 the
// ping-pong one (two groups of waves half a period apart), 0 the default (warp-specialised: layer-1 waves / layer-2 waves)
extern "C" void sgv3d_centerhead_bf16_select_plain(int plain) { g_head_plain = plain; }
// tools only: device buffer of 4 * nb + 2 int64 that receives cycle-counter stamps of workgroup 0 (NULL switches it off)
extern "C" void sgv3d_centerhead_bf16_debug_stamps(void *buf) { g_head_dbg = static_cast<long long *>(buf); }

extern "C" size_t sgv3d_centerhead_bf16_weight_bytes(int num_branches) {
    return num_branches > 0 ? (size_t)num_branches * kW1PerBranch * 2 : 0;
}

extern "C" int sgv3d_centerhead_bf16_pack_weight(const float *w1, int num_branches, int cin, void *w1_packed, void *stream) {
    SGV3D_REQUIRE(w1 && w1_packed && num_branches > 0, "centerhead_bf16_pack_weight: bad argument");
    SGV3D_REQUIRE(cin == kCH, "centerhead_bf16_pack_weight: the bf16 head kernel is built for %d input channels (got %d)", kCH, cin);
    hipLaunchKernelGGL(head_bf16_pack_w1_kernel, dim3(num_branches * 72), dim3(64), 0, as_stream(stream), w1, num_branches,
                       static_cast<__bf16 *>(w1_packed));
    return check_launch("head_bf16_pack_w1_kernel");
}

extern "C" size_t sgv3d_centerhead_bf16_weight2_bytes(int num_branches) {
    return num_branches > 0 ? (size_t)num_branches * kW2Elems * 2 : 0;
}

extern "C" int sgv3d_centerhead_bf16_pack_weight2(const float *w2, const int32_t *out_begin, int num_branches, void *w2_packed,
                                                  void *stream) {
    SGV3D_REQUIRE(w2 && out_begin && w2_packed && num_branches > 0, "centerhead_bf16_pack_weight2: bad argument");
    hipLaunchKernelGGL(head_bf16_pack_w2_kernel, dim3(num_branches), dim3(256), 0, as_stream(stream), w2, out_begin, num_branches,
                       static_cast<__bf16 *>(w2_packed));
    return check_launch("head_bf16_pack_w2_kernel");
}

static int head_bf16_launch(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x, int x_is_bf16,
                            int num_branches, const void *w1_packed, const float *scale1,
                            const float *shift1, int total_out, const void *w2_packed,
                            const float *bias2, const int32_t *out_begin, float *out, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && num_branches > 0 && total_out > 0, "centerhead_branches_forward_bf16: bad shape");
    SGV3D_REQUIRE(cin == kCH, "centerhead_branches_forward_bf16: built for %d input channels (got %d)", kCH, cin);
    SGV3D_REQUIRE(x_ld >= x_coff + cin && x_ld % 4 == 0 && x_coff % 4 == 0, "centerhead_branches_forward_bf16: bad channel slice");
    SGV3D_REQUIRE(!x_is_bf16 || (x_ld % 8 == 0 && x_coff % 8 == 0), "centerhead_branches_forward_bf16: bf16 input needs x_ld / x_coff multiples of 8");
    SGV3D_REQUIRE(x && w1_packed && scale1 && shift1 && w2_packed && bias2 && out_begin && out, "centerhead_branches_forward_bf16: null pointer");
    SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w1_packed) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(w2_packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(scale1) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(shift1) & 15) == 0,
                  "centerhead_branches_forward_bf16: x, scale1, shift1 and the packed weights must be 16-B aligned");
    SGV3D_REQUIRE(batch <= 65535, "centerhead_branches_forward_bf16: batch exceeds grid.y");
    SGV3D_REQUIRE(num_branches <= kMaxBranches, "centerhead_branches_forward_bf16: at most %d branches (got %d)", kMaxBranches, num_branches);
    HeadBf16Args a;
    a.x = x; a.w1 = static_cast<const __bf16 *>(w1_packed); a.scale1 = scale1; a.shift1 = shift1; a.w2 = static_cast<const __bf16 *>(w2_packed); a.bias2 = bias2;
    a.out_begin = out_begin; a.out = out; a.H = h; a.W = w; a.x_ld = x_ld; a.x_coff = x_coff; a.nb = num_branches;
    a.total_out = total_out; a.tiles_x = cdiv(w, kTP); a.tiles_y = cdiv(h, kTP);
    a.dbg = g_head_dbg;
    a.x_bf16 = x_is_bf16 ? 1 : 0;
    if (g_head_dbg == nullptr && g_head_plain == 3) {   // the ping-pong kernel (two groups of waves half a period apart)
        static PerDeviceSize lds_pp;
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&head_bf16_pp_kernel), kWsLds, lds_pp))
            return fail(SGV3D_ELAUNCH, "centerhead_branches_forward_bf16: cannot raise the dynamic LDS limit to %d", kWsLds);
        hipLaunchKernelGGL(head_bf16_pp_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(512), kWsLds, as_stream(stream), a);
        return check_launch("head_bf16_pp_kernel");
    }
    if (g_head_dbg == nullptr && g_head_plain == 0) {   // default: the warp-specialised kernel (layer 2 overlapped with layer 1)
        static PerDeviceSize lds_ws;
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&head_bf16_ws_kernel), kWsLds, lds_ws))
            return fail(SGV3D_ELAUNCH, "centerhead_branches_forward_bf16: cannot raise the dynamic LDS limit to %d", kWsLds);
        hipLaunchKernelGGL(head_bf16_ws_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(512), kWsLds, as_stream(stream), a);
        return check_launch("head_bf16_ws_kernel");
    }
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&head_bf16_kernel), kHeadLds, lds_set))
        return fail(SGV3D_ELAUNCH, "centerhead_branches_forward_bf16: cannot raise the dynamic LDS limit to %d", kHeadLds);
    hipLaunchKernelGGL(head_bf16_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(256), kHeadLds, as_stream(stream), a);
    return check_launch("head_bf16_kernel");
}

extern "C" int sgv3d_centerhead_branches_forward_bf16(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                                      int num_branches, const void *w1_packed, const float *scale1,
                                                      const float *shift1, int total_out, const void *w2_packed,
                                                      const float *bias2, const int32_t *out_begin, float *out, void *stream) {
    return head_bf16_launch(batch, h, w, cin, x_ld, x_coff, x, 0, num_branches, w1_packed, scale1, shift1, total_out, w2_packed,
                            bias2, out_begin, out, stream);
}

extern "C" int sgv3d_centerhead_branches_forward_bf16x(int batch, int h, int w, int cin, int x_ld, int x_coff, const void *x_bf16,
                                                       int num_branches, const void *w1_packed, const float *scale1,
                                                       const float *shift1, int total_out, const void *w2_packed,
                                                       const float *bias2, const int32_t *out_begin, float *out, void *stream) {
    return head_bf16_launch(batch, h, w, cin, x_ld, x_coff, static_cast<const float *>(x_bf16), 1, num_branches, w1_packed, scale1,
                            shift1, total_out, w2_packed, bias2, out_begin, out, stream);
}
