// 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores with the input patch resident in LDS ("patch kernel"),
// the bf16-mode algorithm for the layers Winograd covers in fp32: HeightNet / MSCThead 3x3 blocks, BEV trunk, ResNet 3x3s
// (reference call sites: layers/backbones/lss_fpn.py:166-198, bsm_lss_fpn.py:185-257, layers/heads/bev_height_head.py:75-110
// through mmdet's BasicBlock / Bottleneck).
//
// The implicit-GEMM bf16 kernel (conv_igemm.hip) re-stages every operand tile through registers and LDS once per 32 k and
// synchronises the workgroup every 8 MFMAs per wave; it runs these layers at 0.45-0.65 PFLOP/s.  Here a workgroup owns a
// 16 x 32 pixel tile and 64 output channels:
//   * the (18 x 34) input patch of 32 channels lives in LDS as bf16, laid out [row][8-channel chunk][x][8]: the 32 lanes of a
//     pixel fragment read are consecutive pixels of one row = contiguous 512 bytes (conflict-free, no swizzle) and the tap /
//     k-step part of every address is an instruction immediate; the next 32 channels are staged into the other buffer while
//     the current ones are multiplied -- ONE barrier per 32 input channels = per 144 MFMAs of a wave;
//   * the weights are packed on the host in MFMA-fragment order and stream from L2 straight into registers (16-byte loads,
//     one tap ahead, shared by the four waves through L1), never touching LDS;
//   * a wave owns 4 rows x 32 pixels x 64 channels = 8 accumulator tiles; the product is computed transposed
//     (C^T = W . X^T), so the epilogue (folded BN, residual, ReLU in fp32) goes through a per-wave LDS stage and leaves as
//     16-byte stores of 8 channels (bf16 output) or 4 channels (f32 output).
// Bound: MFMA bf16 (2.5 PFLOP/s dense); algorithmic work 2 * B*H*W * cout * 9 * cin per launch.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kTY = 16, kTX = 32;                // output tile
constexpr int kPY = kTY + 2, kPX = kTX + 2;      // patch
constexpr int kCK = 32;                          // input channels per stage (4 chunks of 8)
constexpr int kChunkB = kPX * 16;                // 544 bytes: one 8-channel chunk of one patch row
constexpr int kRowB = 4 * kChunkB;               // 2176
constexpr int kBufB = kPY * kRowB;               // 39 168 bytes per buffer
constexpr int kSlots = kPY * 4 * kPX;            // 2448 16-byte slots per stage
constexpr int kStageLd = 36;                     // floats per staged pixel row in the epilogue
constexpr int kPatchLds = 2 * kBufB;             // 78 336 (the epilogue stage, 4 x 32 x 36 floats, reuses it)
constexpr int kWFrag = 64 * 8;                   // bf16 elements of one fragment (64 lanes x 8)

// weights: OIHW f32 -> [cout tile of 64][cin chunk of 32][tap][k-step of 16][n-tile of 32][lane][8] bf16, zero beyond cout
__global__ __launch_bounds__(64) void patch_pack_kernel(const float *__restrict__ w, int cout, int cin, __bf16 *__restrict__ out) {
    const int id = blockIdx.x;                     // (((ct * nchunk + ck) * 9 + tap) * 2 + ks) * 2 + nt
    const int nchunk = cin / kCK;
    const int nt = id & 1, ks = (id >> 1) & 1;
    const int tap = (id >> 2) % 9;
    const int ck = ((id >> 2) / 9) % nchunk;
    const int ct = (id >> 2) / 9 / nchunk;
    const int l = threadIdx.x;
    const int co = ct * 64 + nt * 32 + (l & 31);
    __bf16 *dst = out + (size_t)id * kWFrag + l * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ck * kCK + ks * 16 + 8 * (l >> 5) + j;
        dst[j] = co < cout ? (__bf16)w[((size_t)co * cin + ci) * 9 + tap] : (__bf16)0.f;
    }
}

struct PatchArgs {
    const void *x;             // NHWC [B, H, W, x_ld], f32 or bf16
    const __bf16 *w;           // packed fragments
    const float *scale, *bias; // folded BN / bias per output channel (may be NULL)
    const void *res;           // residual NHWC [B, H, W, res_ld] in the OUTPUT dtype, or NULL
    void *y;                   // NHWC [B, H, W, y_ld], f32 or bf16
    int H, W, cin, cout, x_ld, x_coff, y_ld, y_coff, res_ld, relu, tiles_x, tiles_y, ctiles;
    int split;                 // > 1: blockIdx.z owns a range of the 32-channel stages and stores raw partial sums to ws
    float *ws;                 // [split][B*H*W][cout] (the layout of the implicit-GEMM split-K, reduced by the same kernel)
    long long *dbg;            // tools only: cycle-counter stamps of workgroup 0 (2 + 2 * stages + 1 values), else NULL
};

template <bool XB, bool YB>
__global__ __launch_bounds__(256, 2) void conv_patch_bf16_kernel(const PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x / a.ctiles, ct = blockIdx.x - tile * a.ctiles;
    const int b = blockIdx.y;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int y0 = ty * kTY, x0 = tx * kTX;
    const int nchunk = a.cin / kCK;
    const size_t img = (size_t)b * a.H * a.W;
    const int ck0 = a.split > 1 ? (int)((long long)nchunk * blockIdx.z / a.split) : 0;
    const int ck1 = a.split > 1 ? (int)((long long)nchunk * (blockIdx.z + 1) / a.split) : nchunk;

#define PATCH_STAMP(slot) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) a.dbg[slot] = (long long)__builtin_readcyclecounter(); } while (0)
    PATCH_STAMP(0);
    // ---- staging: this thread's (up to 10) 16-byte slots of a 32-channel patch stage ---------------------------------------
    constexpr int kPer = (kSlots + 255) / 256;      // 10
    int s_lds[kPer];
    int s_src[kPer];                                // element offset (within image b) of the slot's 8 channels at chunk 0, or -1 (outside)
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int e = tid + i * 256;                // e = (row * kPX + px) * 4 + c8 : a pixel's 4 chunks on 4 consecutive lanes
        s_lds[i] = -1;
        s_src[i] = -1;
        if (e < kSlots) {
            const int c8 = e & 3, p = e >> 2;
            const int py = p / kPX, px = p - py * kPX;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            s_lds[i] = py * kRowB + c8 * kChunkB + px * 16;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) s_src[i] = (gy * a.W + gx) * a.x_ld + a.x_coff + c8 * 8;
        }
    }
    bf16x8 sreg[kPer];
    auto stage_load = [&](int ck) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
            if (s_src[i] >= 0) {
                if constexpr (XB) {
                    v = *reinterpret_cast<const bf16x8 *>(static_cast<const __bf16 *>(a.x) + img * a.x_ld + s_src[i] + ck * kCK);
                } else {
                    const float *src = static_cast<const float *>(a.x) + img * a.x_ld + s_src[i] + ck * kCK;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(src), hi = *reinterpret_cast<const f32x4 *>(src + 4);
                    const bf16x4 l4 = __builtin_convertvector(lo, bf16x4), h4 = __builtin_convertvector(hi, bf16x4);
                    v = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            sreg[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (s_lds[i] >= 0) *reinterpret_cast<bf16x8 *>(smem + buf * kBufB + s_lds[i]) = sreg[i];
    };

    // ---- compute geometry: wave = rows 4 wave .. 4 wave + 3, m-tile = one row of 32 pixels, lane = pixel x ----------------
    const int abase = (wave * 4) * kRowB + h * kChunkB + r * 16;          // + buf, + (mt + ky) kRowB + kx 16 + ks 2 kChunkB
    const bf16x8 *wl = reinterpret_cast<const bf16x8 *>(a.w) + (size_t)ct * nchunk * 9 * 4 * 64 + lane;   // fragment (ck, tap, ks, nt)
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    stage_load(ck0);
    stage_store(ck0 & 1);
    bf16x8 bq[2][2][2];                 // weight fragments of the current / next tap: [buffer][k-step][n-tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = wl[(size_t)ck0 * 9 * 4 * 64 + (size_t)(ks * 2 + nt) * 64];
    __syncthreads();
    PATCH_STAMP(1);

    for (int ck = ck0; ck < ck1; ++ck) {
        const int buf = ck & 1;
        const char *pa = smem + buf * kBufB + abase;
        const bool more = ck + 1 < ck1;
        if (more) stage_load(ck + 1);                                    // global loads of the next stage fly under the MFMAs
        const size_t wbase = (size_t)ck * 9 * 4 * 64;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1, nxt = cur ^ 1;
            {
                // fragments of the next tap (of the next chunk's first tap after tap 8; harmless re-read at the very end)
                const size_t nb = tap < 8 ? wbase + (size_t)(tap + 1) * 4 * 64 : (more ? wbase + (size_t)9 * 4 * 64 : wbase);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq[nxt][ks][nt] = wl[nb + (size_t)(ks * 2 + nt) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int toff = (tap / 3) * kRowB + (tap % 3) * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const bf16x8 *>(pa + mt * kRowB + toff + ks * 2 * kChunkB);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[cur][ks][nt], af[mt], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // 9 taps: tap 8 used buffer 0 and filled buffer 1 with the next chunk's tap 0 -> make it buffer 0
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[0][ks][nt] = bq[1][ks][nt];
        PATCH_STAMP(2 + 2 * (ck - ck0));
        if (more) stage_store(buf ^ 1);            // its readers (chunk ck - 1) passed the barrier that ended the previous iteration
        __syncthreads();
        PATCH_STAMP(3 + 2 * (ck - ck0));
    }

    // ---- epilogue: transposed accumulators (pixel on the lane, 4 consecutive channels in registers 4 g .. 4 g + 3) -> per-wave
    // LDS stage -> lane = (pixel lane >> 2 (+16), 8-channel chunk lane & 3) -> folded BN, residual, ReLU -> 16- / 32-byte rows
    float *stage = reinterpret_cast<float *>(smem) + wave * (32 * kStageLd);
    const int pc = lane & 3, pp = lane >> 2;
    // per-channel terms and (no split) the residual rows of all 8 tiles of the wave first: one memory latency in all,
    // not one per tile behind the previous tile's stores
    f32x4 sc0_[2], sc1_[2], sh0_[2], sh1_[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ch = ct * 64 + nt * 32 + 8 * pc;
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
        sc0_[nt] = sc1_[nt] = one;
        sh0_[nt] = sh1_[nt] = zero;
        if (ch < a.cout && a.scale) { sc0_[nt] = *reinterpret_cast<const f32x4 *>(a.scale + ch); sc1_[nt] = *reinterpret_cast<const f32x4 *>(a.scale + ch + 4); }
        if (ch < a.cout && a.bias) { sh0_[nt] = *reinterpret_cast<const f32x4 *>(a.bias + ch); sh1_[nt] = *reinterpret_cast<const f32x4 *>(a.bias + ch + 4); }
    }
    bf16x8 resq[2][4][2];             // bf16 output: the residual of the 8 tiles (an f32 residual is read at its use: 128 registers)
    const bool has_res = a.res != nullptr && a.split <= 1;
    if (YB && has_res) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int ch = ct * 64 + nt * 32 + 8 * pc;
                    const int gy = y0 + wave * 4 + mt, gx = x0 + pp + 16 * ps;
                    if (gy < a.H && gx < a.W && ch < a.cout) {
                        const size_t pix = img + (size_t)gy * a.W + gx;
                        resq[nt][mt][ps] = *reinterpret_cast<const bf16x8 *>(static_cast<const __bf16 *>(a.res) + pix * a.res_ld + ch);
                    }
                }
    }
    PATCH_STAMP(3 + 2 * (ck1 - ck0));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (nt == 1) PATCH_STAMP(4 + 2 * (ck1 - ck0));
        const int ch = ct * 64 + nt * 32 + 8 * pc;
        const bool ch_ok = ch < a.cout;
        const f32x4 sc0 = sc0_[nt], sc1 = sc1_[nt], sh0 = sh0_[nt], sh1 = sh1_[nt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stage + r * kStageLd + 8 * g + 4 * h) = v;
            }
            __builtin_amdgcn_wave_barrier();
            const int gy = y0 + wave * 4 + mt;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int p = pp + 16 * ps;
                const int gx = x0 + p;
                f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc);
                f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc + 4);
                if (a.split > 1) {
                    if (gy < a.H && gx < a.W && ch_ok) {
                        float *wp = a.ws + ((size_t)blockIdx.z * gridDim.y * a.H * a.W + img + (size_t)gy * a.W + gx) * a.cout + ch;
                        *reinterpret_cast<f32x4 *>(wp) = v0;
                        *reinterpret_cast<f32x4 *>(wp + 4) = v1;
                    }
                } else if (gy < a.H && gx < a.W && ch_ok) {
                    const size_t pix = img + (size_t)gy * a.W + gx;
                    v0 = v0 * sc0 + sh0;
                    v1 = v1 * sc1 + sh1;
                    if (has_res) {
                        if constexpr (YB) {
                            const bf16x8 rq = resq[nt][mt][ps];
#pragma unroll
                            for (int i = 0; i < 4; ++i) { v0[i] += (float)rq[i]; v1[i] += (float)rq[4 + i]; }
                        } else {
                            const float *rp = static_cast<const float *>(a.res) + pix * a.res_ld + ch;
                            v0 += *reinterpret_cast<const f32x4 *>(rp);
                            v1 += *reinterpret_cast<const f32x4 *>(rp + 4);
                        }
                    }
                    if (a.relu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], 0.f); v1[i] = fmaxf(v1[i], 0.f); }
                    }
                    if constexpr (YB) {
                        const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
                        *reinterpret_cast<bf16x8 *>(static_cast<__bf16 *>(a.y) + pix * a.y_ld + a.y_coff + ch) =
                            __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7);
                    } else {
                        float *yp = static_cast<float *>(a.y) + pix * a.y_ld + a.y_coff + ch;
                        *reinterpret_cast<f32x4 *>(yp) = v0;
                        *reinterpret_cast<f32x4 *>(yp + 4) = v1;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    PATCH_STAMP(2 + 2 * (ck1 - ck0));
#undef PATCH_STAMP
}

static long long *g_patch_dbg = nullptr;

template <bool XB, bool YB>
int launch_patch(const PatchArgs &a, int batch, hipStream_t st) {
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_patch_bf16_kernel<XB, YB>), kPatchLds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv3x3_patch_bf16: cannot raise the dynamic LDS limit to %d", kPatchLds);
    hipLaunchKernelGGL((conv_patch_bf16_kernel<XB, YB>), dim3(a.tiles_x * a.tiles_y * a.ctiles, batch, a.split), dim3(256), kPatchLds, st, a);
    if (a.split > 1) {                   // second stage shared with the implicit GEMM: fixed-order sum of the partials + epilogue
        ConvArgs r = {};
        r.scale = a.scale; r.bias = a.bias; r.res = static_cast<const float *>(a.res); r.y = static_cast<float *>(a.y);
        r.M = batch * a.H * a.W; r.N = a.cout; r.cout = a.cout; r.m_h = a.H; r.m_w = a.W; r.out_h = a.H; r.out_w = a.W;
        r.y_ld = a.y_ld; r.y_coff = a.y_coff; r.res_ld = a.res_ld; r.relu = a.relu;
        r.mode = SGV3D_CONV_NORMAL | (YB ? kConvYBf16 | kConvResBf16 : 0);
        r.split_k = a.split; r.ws = a.ws;
        return launch_splitk_reduce(r, st);
    }
    return check_launch("conv_patch_bf16_kernel");
}

}  // namespace

extern "C" void sgv3d_conv3x3_patch_bf16_debug_stamps(void *buf) { g_patch_dbg = static_cast<long long *>(buf); }

extern "C" size_t sgv3d_conv3x3_patch_bf16_weight_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % kCK) return 0;
    return (size_t)cdiv(cout, 64) * (cin / kCK) * 9 * 4 * kWFrag * 2;
}

extern "C" int sgv3d_conv3x3_patch_bf16_pack_weight(const float *w, int cout, int cin, void *w_packed, void *stream) {
    SGV3D_REQUIRE(w && w_packed && cout > 0 && cin > 0 && cin % kCK == 0, "conv3x3_patch_bf16_pack_weight: cin must be a multiple of %d", kCK);
    const int frags = cdiv(cout, 64) * (cin / kCK) * 9 * 4;
    hipLaunchKernelGGL(patch_pack_kernel, dim3(frags), dim3(64), 0, as_stream(stream), w, cout, cin, static_cast<__bf16 *>(w_packed));
    return check_launch("patch_pack_kernel");
}

extern "C" int sgv3d_conv3x3_patch_bf16_forward(int batch, int h, int w, int cin, int cout, int x_ld, int x_coff, int y_ld,
                                                int y_coff, int res_ld, int relu, const void *x, const void *w_packed,
                                                const float *scale, const float *bias, const void *residual, void *y,
                                                int io_flags, int split_k, void *workspace, size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv3x3_patch_bf16: bad shape");
    SGV3D_REQUIRE(cin % kCK == 0 && cout % 8 == 0, "conv3x3_patch_bf16: cin must be a multiple of 32 and cout of 8 (got %d / %d)", cin, cout);
    SGV3D_REQUIRE(x && w_packed && y, "conv3x3_patch_bf16: null pointer");
    SGV3D_REQUIRE(x_ld >= x_coff + cin && y_ld >= y_coff + cout && (residual == nullptr || res_ld >= cout), "conv3x3_patch_bf16: channel strides too small");
    SGV3D_REQUIRE(x_ld % 8 == 0 && x_coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0 && (residual == nullptr || res_ld % 8 == 0),
                  "conv3x3_patch_bf16: channel strides / offsets must be multiples of 8");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "conv3x3_patch_bf16: pointers must be 16-B aligned");
    if (split_k < 1) split_k = 1;
    SGV3D_REQUIRE(split_k <= cin / kCK && split_k <= 64, "conv3x3_patch_bf16: split_k = %d exceeds the %d stages of 32 input channels", split_k, cin / kCK);
    if (split_k > 1) {
        SGV3D_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "conv3x3_patch_bf16: split_k needs a 16-B aligned workspace");
        if (workspace_bytes < sizeof(float) * (size_t)split_k * batch * h * w * cout)
            return fail(SGV3D_ENOSPACE, "conv3x3_patch_bf16: workspace too small for split_k = %d", split_k);
    }
    PatchArgs a;
    a.dbg = g_patch_dbg;
    a.split = split_k; a.ws = static_cast<float *>(workspace);
    a.x = x; a.w = static_cast<const __bf16 *>(w_packed); a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.H = h; a.W = w; a.cin = cin; a.cout = cout; a.x_ld = x_ld; a.x_coff = x_coff; a.y_ld = y_ld; a.y_coff = y_coff;
    a.res_ld = res_ld; a.relu = relu; a.tiles_x = cdiv(w, kTX); a.tiles_y = cdiv(h, kTY); a.ctiles = cdiv(cout, 64);
    hipStream_t st = as_stream(stream);
    switch (io_flags & 3) {
        case 0: return launch_patch<false, false>(a, batch, st);
        case 1: return launch_patch<true, false>(a, batch, st);
        case 2: return launch_patch<false, true>(a, batch, st);
        default: return launch_patch<true, true>(a, batch, st);
    }
}
