// Pointwise (1x1) convolution with bf16 activations in HBM on the bf16 matrix cores ("pointwise kernel"): the reducing /
// expanding 1x1 layers and the strided 1x1 shortcuts of every ResNet bottleneck in bf16 mode (reference call sites: mmdet's
// Bottleneck through layers/backbones/lss_fpn.py:296-301 `self.img_backbone`, exps/.../bev_height_lss_r101_*.py backbone
// dicts).  A quarter of a cfg-3 step is spent in these layers, and most of them are bound by HBM, not by the matrix pipe
// (256 -> 1024 at 4 x 68 x 120 with residual: 150 MB for 17 GFLOP).  The implicit-GEMM bf16 kernel (conv_igemm.hip) runs
// them at 2-3 x their HBM time: both operand tiles pass through registers and ds_write every 32 k (the LDS store path is a
// third of the read rate), a workgroup synchronises every 8 MFMAs per wave, and at 220 VGPRs only two workgroups share a CU,
// so a workgroup's residual / store phase has nothing to overlap with.  Here:
//   * the weights never touch LDS: they are packed on the host in MFMA-fragment order ([n-tile of 32][k-step of 16][lane][8])
//     and stream from L2 straight into a ring of fragment registers one 64-k chunk ahead, 1 KB contiguous per wave load with a
//     scalar offset (no vector instruction per load);
//   * only the activation rows go through LDS: 64 k (128 B) per row and chunk, 16-byte loads and ds_write_b128, double
//     buffered, one barrier per chunk = per 16 MFMAs of a wave; 16-byte slots XOR-swizzled with the row so that both the
//     fragment reads and the stores are conflict-free without padding;
//   * a wave owns 64 pixels x 64 channels (64 accumulator registers): ~128-168 VGPRs, 18-64 KB of LDS -- three workgroups per
//     CU, whose load, MFMA and store phases overlap each other;
//   * the residual rows are requested before the last chunk and arrive under its MFMAs;
//   * the product is computed transposed (C^T = W . X^T: pixel on the lane, 4 consecutive channels in a register quad); the
//     epilogue (folded BN, residual, ReLU in f32, one rounding) goes through a per-wave LDS stage and leaves as 16-byte stores
//     of 8 channels: 4 lanes write one 64-byte row segment.
// Workgroup tile (64 WM) pixels x (64 WN) channels with WM x WN = 4 waves: 64 x 256 (expanding layers), 128 x 128, 256 x 64
// (reducing layers with 64 outputs).  The grid walks the channel tiles of one pixel tile back to back on one XCD, so the
// input rows come from HBM once.
// Bound: HBM for the large maps (2 * (M * K + M * N [+ M * N residual]) bytes), MFMA bf16 (2 * M * N * K flop) for the deep ones.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kKC = 64;                 // k per chunk: one 128-byte row segment
constexpr int kRowB = kKC * 2;          // bytes per staged row
constexpr int kStageLd = 36;            // floats per staged pixel row in the epilogue (144 B: conflict-free b128 rows)
constexpr int kFragB = 64 * 16;         // bytes of one weight fragment (64 lanes x 8 bf16)

struct PwArgs {
    const void *x;             // NHWC bf16 [B, in_h, in_w, x_ld]
    const void *w;             // packed fragments (pw_pack_kernel)
    const float *scale, *bias; // folded BN / bias per output channel (may be NULL)
    const void *res;           // bf16 residual [M, res_ld] or NULL
    void *y;                   // bf16 [M, y_ld]
    int M, N, K;
    int x_ld, x_coff, y_ld, y_coff, res_ld, relu;
    int in_h, in_w, out_h, out_w, stride;
    int tiles_m, tiles_n, nch, ksteps;   // nch = ceil(K / 64), ksteps = 4 nch (k-steps of 16 per n-tile in the packed weights)
    unsigned x_bytes, w_bytes, res_bytes;
};

// weights: OIHW f32 [cout][cin][1][1] -> [n-tile of 32][k-step of 16][lane][8] bf16; lane l holds channel 32 nt + (l & 31),
// k = 16 ks + 8 (l >> 5) .. + 8 (the operand layout of v_mfma_f32_32x32x16_bf16); zero beyond cout / cin
__global__ __launch_bounds__(64) void pw_pack_kernel(const float *__restrict__ w, int cout, int cin, int ksteps, __bf16 *__restrict__ out) {
    const int id = blockIdx.x;                     // nt * ksteps + ks
    const int nt = id / ksteps, ks = id - nt * ksteps;
    const int l = threadIdx.x;
    const int co = nt * 32 + (l & 31);
    __bf16 *dst = out + ((size_t)id * 64 + l) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ks * 16 + 8 * (l >> 5) + j;
        dst[j] = (co < cout && ci < cin) ? (__bf16)w[(size_t)co * cin + ci] : (__bf16)0.f;
    }
}

template <int WM, int WN>
__global__ __launch_bounds__(256, WM == 4 ? 2 : 3) void conv_pw_bf16_kernel(const PwArgs a) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_LD = BM / 32;                            // 16-byte loads per thread and chunk
    constexpr int kBufB = BM * kRowB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = __builtin_amdgcn_readfirstlane(WN == 1 ? wave : WN == 2 ? wave >> 1 : 0);
    const int wn = __builtin_amdgcn_readfirstlane(WN == 1 ? 0 : WN == 2 ? wave & 1 : wave);
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware walk: the workgroups that share an XCD's L2 take consecutive logical tiles; the channel tile changes fastest
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = (int)((unsigned)logical / (unsigned)a.tiles_n);
    const int tn = logical - tm * a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.w_bytes, 0x00020000);

    // activation rows: thread -> (16-byte chunk c8 of the 128-byte row segment, rows r0 + 32 i)
    const int c8 = tid & 7, r0 = tid >> 3;
    unsigned a_off[A_LD];
    const bool pointwise = a.stride == 1;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + r0 + 32 * i;
        long long pix = m;
        if (!pointwise) {
            const int mm = m < a.M ? m : 0;
            const int t = (int)((unsigned)mm / (unsigned)a.out_w), ow = mm - t * a.out_w;
            const int img = (int)((unsigned)t / (unsigned)a.out_h), oh = t - img * a.out_h;
            pix = ((long long)img * a.in_h + oh * a.stride) * a.in_w + ow * a.stride;
        }
        a_off[i] = m < a.M ? (unsigned)((pix * a.x_ld + a.x_coff + c8 * 8) * 2) : 0xffffffffu;
    }
    // K % 64 == 32: the upper half of the last chunk lies beyond K -- those lanes read zeros (not the neighbouring channels)
    const bool tail_dead = (a.K & 63) != 0 && c8 >= 4;
    const int st_slot = (c8 ^ ((r0 >> 1) & 7)) * 16;                 // rows r0 + 32 i share (row >> 1) & 7
    char *const st_ptr = smem + r0 * kRowB + st_slot;

    // fragment reads: row = wm 64 + mt 32 + lr, logical 16-byte chunk 2 s + lh of k-step s
    const int swz = (lr >> 1) & 7;
    int rd_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rd_off[s] = (wm * 64 + lr) * kRowB + (((2 * s + lh) ^ swz) * 16);

    // weight fragments of this wave's two n-tiles: byte offset = ((nt * ksteps + ks) * 64 + lane) * 16
    // (the packed weights hold an even number of n-tiles; a wave whose 64 channels lie beyond N reads zeros)
    const unsigned w_lane = n0 + wn * 64 < a.N ? lane * 16 : 0xffffffffu;
    const int nt0 = (n0 + wn * 64) >> 5;
    const int w_s0 = __builtin_amdgcn_readfirstlane(nt0 * a.ksteps * kFragB);
    const int w_s1 = __builtin_amdgcn_readfirstlane((nt0 + 1) * a.ksteps * kFragB);

    u32x4 ra[A_LD];
    bf16x8 wf[4][2];
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

// (C may lie beyond the last chunk: the request is then out of range for every lane and returns zeros without touching memory --
// the steady-state loop stays branch-free, which keeps the compiler's vmcnt counts exact)
#define PW_LOAD_A(C)                                                                                       \
    do {                                                                                                   \
        const bool dead_ = (C) >= a.nch || (tail_dead && (C) == a.nch - 1);                                \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                   \
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead_ ? 0xffffffffu : a_off[i], (C) * kRowB, 0)); \
    } while (0)
#define PW_STORE_A(BUF)                                                                                    \
    do {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                   \
            *reinterpret_cast<u32x4 *>(st_ptr + (BUF) * kBufB + i * 32 * kRowB) = ra[i];                   \
    } while (0)
#define PW_LOAD_W(C, S)                                                                                    \
    do {                                                                                                   \
        wf[S][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, w_s0 + ((C) * 4 + (S)) * kFragB, 0)); \
        wf[S][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, w_s1 + ((C) * 4 + (S)) * kFragB, 0)); \
    } while (0)
#define PW_MFMA_STEP(BUF, S)                                                                               \
    do {                                                                                                   \
        const bf16x8 fa0 = *reinterpret_cast<const bf16x8 *>(smem + (BUF) * kBufB + rd_off[S]);            \
        const bf16x8 fa1 = *reinterpret_cast<const bf16x8 *>(smem + (BUF) * kBufB + rd_off[S] + 32 * kRowB); \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][0], fa0, acc[0][0], 0, 0, 0);            \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][1], fa0, acc[0][1], 0, 0, 0);            \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][0], fa1, acc[1][0], 0, 0, 0);            \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][1], fa1, acc[1][1], 0, 0, 0);            \
    } while (0)

    // residual rows in the epilogue's layout: lane -> (pixel (lane >> 2) + 16 ps, 8-channel chunk lane & 3) of a 32 x 32 tile;
    // buffer loads with 32-bit offsets (rows beyond M / chunks beyond N: out of range, zeros)
    const int pc = lane & 3, pp = lane >> 2;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, a.res ? (int)a.res_bytes : 0, 0x00020000);
    bf16x8 resq[2][2][2];
#define PW_FETCH_RES()                                                                                     \
    do {                                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
            _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                             \
                const int row = m0 + wm * 64 + mt * 32 + pp + 16 * ps;                                     \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                         \
                    const int ch = n0 + wn * 64 + nt * 32 + 8 * pc;                                        \
                    const unsigned ro = (row < a.M && ch < a.N) ? ((unsigned)row * (unsigned)a.res_ld + ch) * 2u : 0xffffffffu; \
                    resq[mt][nt][ps] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, ro, 0, 0)); \
                }                                                                                          \
            }                                                                                              \
    } while (0)

    // The order of the memory instructions is pinned with sched_barriers: left alone, the compiler sinks every load of an
    // iteration behind its MFMAs (their destination registers double as LDS fragment registers) and then waits for all of
    // them at the top of the next one -- a full memory latency per chunk.
#define PW_SB() __builtin_amdgcn_sched_barrier(0)
    // Prologue in the loop's order -- activation rows first, then the fragments -- so that the wait in front of the loop's
    // ds_write counts the same 8 younger fragment loads on the first pass as on every other (chunk 0 in registers of its own).
    {
        u32x4 rp[A_LD];
        const bool dead_ = tail_dead && a.nch == 1;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            rp[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead_ ? 0xffffffffu : a_off[i], 0, 0));
        PW_LOAD_A(1);
        PW_SB();
#pragma unroll
        for (int s = 0; s < 4; ++s) PW_LOAD_W(0, s);
        PW_SB();
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4 *>(st_ptr + i * 32 * kRowB) = rp[i];
    }
    PW_SB();
    __syncthreads();
    const int last = a.nch - 1;
    for (int c = 0; c < last; ++c) {
        const int buf = c & 1;
        PW_STORE_A(buf ^ 1);                            // chunk c + 1, requested one iteration ago (the 8 fragment loads behind it stay in flight)
        PW_SB();
        PW_LOAD_A(c + 2);
        PW_SB();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            PW_MFMA_STEP(buf, s);
            PW_SB();
            PW_LOAD_W(c + 1, s);                        // into the fragment registers this k-step has just used
            PW_SB();
        }
        __syncthreads();
    }
    PW_FETCH_RES();                                     // arrives under the last chunk's MFMAs
    PW_SB();
    {
        const int buf = last & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) PW_MFMA_STEP(buf, s);
    }
    PW_SB();
    __syncthreads();                                    // the epilogue stage reuses the activation buffers
#undef PW_SB
#undef PW_LOAD_A
#undef PW_STORE_A
#undef PW_LOAD_W
#undef PW_FETCH_RES
#undef PW_MFMA_STEP

    // epilogue.  acc[mt][nt][4 g + i] = C[pixel m0 + wm 64 + 32 mt + lr][channel n0 + wn 64 + 32 nt + 8 g + 4 lh + i].
    // The loop ended on a barrier: the activation buffers are dead, each wave stages its tiles in its own 32 x 36 f32 slice.
    float *const stage = reinterpret_cast<float *>(smem) + wave * (32 * kStageLd);
    __bf16 *const yb = reinterpret_cast<__bf16 *>(a.y);
    const int prow0 = m0 + wm * 64, pcol0 = n0 + wn * 64;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ch = pcol0 + nt * 32 + 8 * pc;
        const bool ch_ok = ch < a.N;
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 sc0 = one, sc1 = one, sh0 = zero, sh1 = zero;
        if (ch_ok && a.scale) { sc0 = *reinterpret_cast<const f32x4 *>(a.scale + ch); sc1 = *reinterpret_cast<const f32x4 *>(a.scale + ch + 4); }
        if (ch_ok && a.bias) { sh0 = *reinterpret_cast<const f32x4 *>(a.bias + ch); sh1 = *reinterpret_cast<const f32x4 *>(a.bias + ch + 4); }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stage + lr * kStageLd + 8 * g + 4 * lh) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int p = pp + 16 * ps;
                const int row = prow0 + mt * 32 + p;
                f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc);
                f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc + 4);
                if (row < a.M && ch_ok) {
                    v0 = v0 * sc0 + sh0;
                    v1 = v1 * sc1 + sh1;
                    {                                   // (zeros when there is no residual)
                        const bf16x8 rq = resq[mt][nt][ps];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v0[i] += (float)rq[i]; v1[i] += (float)rq[4 + i]; }
                    }
                    if (a.relu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], 0.f); v1[i] = fmaxf(v1[i], 0.f); }
                    }
                    const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
                    *reinterpret_cast<bf16x8 *>(yb + (size_t)row * a.y_ld + a.y_coff + ch) = __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int WM, int WN>
int launch_pw(const PwArgs &a0, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    PwArgs a = a0;
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, BN);
    constexpr size_t tiles = 2 * (size_t)BM * kRowB, stage = sizeof(float) * 4 * 32 * kStageLd;
    constexpr size_t lds = tiles > stage ? tiles : stage;
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_pw_bf16_kernel<WM, WN>), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv_pw_bf16: cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL((conv_pw_bf16_kernel<WM, WN>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, st, a);
    return check_launch("conv_pw_bf16_kernel");
}

}  // namespace

extern "C" size_t sgv3d_conv_pw_bf16_weight_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)cdiv(cout, 64) * 2 * (cdiv(cin, kKC) * 4) * kFragB;
}

extern "C" int sgv3d_conv_pw_bf16_pack_weight(const float *w, int cout, int cin, void *w_packed, void *stream) {
    SGV3D_REQUIRE(w && w_packed && cout > 0 && cin > 0, "conv_pw_bf16_pack_weight: bad argument");
    const int ksteps = cdiv(cin, kKC) * 4;
    hipLaunchKernelGGL(pw_pack_kernel, dim3(cdiv(cout, 64) * 2 * ksteps), dim3(64), 0, as_stream(stream), w, cout, cin, ksteps,
                       static_cast<__bf16 *>(w_packed));
    return check_launch("pw_pack_kernel");
}

extern "C" int sgv3d_conv_pw_bf16_forward(const sgv3d_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                          const float *bias, const void *residual, void *y, void *stream) {
    SGV3D_REQUIRE(d && x && w_packed && y, "conv_pw_bf16: null pointer");
    SGV3D_REQUIRE(d->kh == 1 && d->kw == 1 && d->pad == 0 && d->dil == 1 && d->stride >= 1 && d->mode == SGV3D_CONV_NORMAL && d->split_k <= 1,
                  "conv_pw_bf16: 1x1 / pad 0 layers in NORMAL mode without split-K only");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0, "conv_pw_bf16: non-positive dimension");
    SGV3D_REQUIRE(d->out_h == (d->in_h - 1) / d->stride + 1 && d->out_w == (d->in_w - 1) / d->stride + 1,
                  "conv_pw_bf16: output %dx%d does not match the conv arithmetic", d->out_h, d->out_w);
    SGV3D_REQUIRE(d->cin % 32 == 0 && d->cout % 8 == 0, "conv_pw_bf16: cin must be a multiple of 32 and cout of 8 (got %d / %d)", d->cin, d->cout);
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout && (residual == nullptr || d->res_ld >= d->cout),
                  "conv_pw_bf16: channel strides too small");
    SGV3D_REQUIRE(d->x_ld % 8 == 0 && d->x_coff % 8 == 0 && d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && (residual == nullptr || d->res_ld % 8 == 0),
                  "conv_pw_bf16: channel strides / offsets must be multiples of 8 (16-byte rows of bf16)");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "conv_pw_bf16: pointers must be 16-B aligned");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    const long long xb = (long long)d->batch * d->in_h * d->in_w * d->x_ld * 2;
    const size_t wb = sgv3d_conv_pw_bf16_weight_bytes(d->cout, d->cin);
    SGV3D_REQUIRE(M < 0x7fffffffLL && xb < 0xf0000000LL && wb < 0xf0000000ULL && (residual == nullptr || M * d->res_ld * 2 < 0xf0000000LL), "conv_pw_bf16: operands larger than 3.75 GiB (32-bit buffer offsets)");
    PwArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.M = (int)M; a.N = d->cout; a.K = d->cin;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld; a.relu = d->relu;
    a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w; a.stride = d->stride;
    a.nch = cdiv(d->cin, kKC);
    a.ksteps = a.nch * 4;
    a.tiles_m = a.tiles_n = 0;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.res_bytes = (unsigned)(M * d->res_ld * 2);
    hipStream_t st = as_stream(stream);
    switch (d->tile) {
        case SGV3D_TILE_PW_64x256: return launch_pw<1, 4>(a, st);
        case SGV3D_TILE_PW_128x128: return launch_pw<2, 2>(a, st);
        case SGV3D_TILE_PW_256x64: return launch_pw<4, 1>(a, st);
        default: return fail(SGV3D_EINVAL, "conv_pw_bf16: desc.tile must be SGV3D_TILE_PW_64x256 / _128x128 / _256x64 (got %d)", d->tile);
    }
}
