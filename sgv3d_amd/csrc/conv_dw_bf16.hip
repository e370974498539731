// Implicit-GEMM convolution with bf16 activations in HBM whose WEIGHTS STREAM FROM L2 IN MFMA-FRAGMENT ORDER ("direct-weight
// kernel"): the bf16-mode algorithm for the 1x1 layers and strided shortcuts of every ResNet bottleneck, the 3x3 layers the
// patch kernel tiles badly (68 x 120 maps), the strided and dilated 3x3 layers (ASPP) and the 7x7 BEV stem (reference call
// sites: mmdet ResNet / SECOND behind layers/backbones/lss_fpn.py:296-301, ASPP / HeightNet lss_fpn.py:58-121,200-259, the BEV
// trunk layers/heads/bev_height_head.py:75-110).  A third of a cfg-3 step is spent in these layers; most of the 1x1 ones are
// bound by HBM (256 -> 1024 at 4 x 68 x 120 with residual: 150 MB for 17 GFLOP), the others by the matrix pipe.
// The implicit-GEMM bf16 kernel (conv_igemm.hip) passes BOTH operand tiles through registers and ds_write every 32 k (the LDS
// store path runs at a third of the read rate), synchronises the workgroup every 8 MFMAs per wave, and at 220 VGPRs only two
// workgroups share a CU, so a workgroup's residual / store phase has nothing to overlap with.  Here:
//   * the weights never touch LDS: packed on the host in MFMA-fragment order ([n-tile of 32][k-step of 16][lane][8], k = (tap,
//     32-channel block); a 64-k chunk is two consecutive blocks, which for cin % 64 == 32 may belong to two taps: no padding) they stream from L2 straight into a ring of fragment registers one 64-k
//     chunk ahead -- 1 KB contiguous per wave load, scalar offset, no vector instruction per load;
//   * only the activation rows go through LDS: 64 k (128 B) per row and chunk gathered with 16-byte buffer loads (padding /
//     out-of-image rows: out-of-range offset, zeros) and ds_write_b128, double buffered, ONE barrier per chunk = per 16-32 MFMAs
//     of a wave; 16-byte slots XOR-swizzled with the row: fragment reads and stores conflict-free without padding;
//   * a wave owns (32 MT) pixels x 64 channels, MT = 2 (64 accumulator registers, <= 128 VGPRs: four workgroups per CU whose
//     load, MFMA and store phases overlap -- the HBM-bound layers) or MT = 4 (twice the MFMAs per weight fragment: the
//     L1-bound deep layers);
//   * the residual rows are requested during the last chunk into the fragment registers that chunk no longer refills;
//   * the product is computed transposed (C^T = W . X^T: pixel on the lane, 4 consecutive channels in a register quad); the
//     epilogue (folded BN, residual, ReLU in f32, one rounding) goes through a per-wave LDS stage and leaves as 16-byte stores
//     of 8 channels: 4 lanes write one 64-byte row segment.
// Workgroup = 4 waves as WM x WN, tile (32 MT WM) pixels x (64 WN) channels.  The grid walks the channel tiles of one pixel
// tile back to back on one XCD, so the input rows come from HBM once.
// Results: f32 accumulation of bf16 products in ascending (tap, channel) order, one rounding on the store.
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

struct DwArgs {
    const void *x;             // NHWC bf16 [B, in_h, in_w, x_ld]
    const void *w;             // packed fragments (dw_pack_kernel)
    const float *scale, *bias; // folded BN / bias per output channel (may be NULL)
    const void *res;           // bf16 residual [M, res_ld] or NULL
    void *y;                   // bf16 [M, y_ld]
    int M, N, cin;
    int x_ld, x_coff, y_ld, y_coff, res_ld, relu;
    int in_h, in_w, out_h, out_w, kh, kw, stride, pad, dil;
    int m_h, m_w;              // pixel grid of the GEMM rows (the output map; the input map of a transposed convolution)
    int ks, cout;              // ks > 0: transposed convolution with kernel == stride == ks as a 1x1 GEMM with N = ks ks cout columns
                               // ordered (dy, dx, co); column (dy, dx, co) of input pixel (ih, iw) lands at output pixel (ih ks + dy, iw ks + dx)
    int tiles_m, tiles_n;
    int bpt, nblk, nch, ksteps;   // 32-channel blocks per tap = cin / 32, nblk = kh kw bpt, nch = ceil(nblk / 2) chunks of 64 k, ksteps = 4 nch
    unsigned x_bytes, w_bytes, res_bytes, y_bytes;
    int split;                 // SPLIT instantiation: workgroups per tile along k (gridDim.y)
    float *ws;                 // [split][M][N] partial sums
    // conv_dw_bf16_pair_kernel only: the 1x1 convolution that follows (K2 = N channels of the first one, N2 outputs)
    const void *w2;            // packed fragments of the second layer
    const float *scale2, *bias2;
    int N2, relu2, ksteps2;
    unsigned w2_bytes;
};

// weights: OIHW f32 [cout][cin_w][kh][kw] -> [n-tile of 32][k-step of 16][lane][8] bf16; lane l holds channel 32 nt + (l & 31),
// k = 16 ks + 8 (l >> 5) .. + 8 (the operand layout of v_mfma_f32_32x32x16_bf16).  k runs over BLOCKS of 32 input channels,
// block = tap * bpt + (channel / 32) with bpt = cin / 32 blocks per tap; a 64-k chunk is two consecutive blocks (for
// cin % 64 == 32 they may belong to two taps: no padding of a tap's channels); zero beyond cout / cin_w / the last block
__global__ __launch_bounds__(64) void dw_pack_kernel(const float *__restrict__ w, int cout, int cin_w, int taps, int bpt, int ksteps,
                                                      __bf16 *__restrict__ out) {
    const int id = blockIdx.x;                     // nt * ksteps + ks
    const int nt = id / ksteps, ks = id - nt * ksteps;
    const int blk = ks >> 1, tap = blk / bpt, cb = blk - tap * bpt;
    const int l = threadIdx.x;
    const int co = nt * 32 + (l & 31);
    __bf16 *dst = out + ((size_t)id * 64 + l) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = cb * 32 + (ks & 1) * 16 + 8 * (l >> 5) + j;
        dst[j] = (co < cout && tap < taps && ci < cin_w) ? (__bf16)w[((size_t)co * cin_w + ci) * taps + tap] : (__bf16)0.f;
    }
}

// NTAIL: N is not a multiple of the tile's 64 WN columns -- some waves (or half-waves) of the last column tile own no channel
// SPLIT: blockIdx.y owns a range of the 64-k chunks and stores raw f32 partial sums to a.ws [split][M][N] (dw_splitk_reduce_kernel adds
// them in split order and runs the epilogue) -- for the maps whose tiles do not fill the chip (cfg-5 at batch 1: 82 - 162 workgroups)
// DEEP: activation rows and weight fragments are requested TWO chunks ahead (two register sets each, the loop unrolled by two) --
// for the layers whose grid leaves one workgroup per CU (cfg-5 at batch 1: 81 tiles of 64 pixels): with one wave per SIMD nothing
// else covers the ~1 - 2 k cycles of an L2 / HBM round trip, and a chunk is only 512 cycles of MFMAs.  Costs 40 registers
// (two workgroups per CU), same arithmetic in the same order: bitwise the plain instantiation.
// NTW = 1: a wave owns ONE 32-channel tile (workgroup = 64 pixels x 128 channels): twice the workgroups on the maps whose 64 x 256
// tiles number fewer than the CUs -- those launches are bound by what one CU's vector-memory path delivers (32 KB of fragments per
// chunk and workgroup at 64 B / clk), and half the fragments per workgroup on twice the CUs is the way to more of those paths.
template <int WM, int WN, int MT, bool NTAIL, bool SPLIT = false, bool DEEP = false, int NTW = 2, int PLAINK = -1>
__global__ __launch_bounds__(256, (MT == 4 || DEEP) ? 2 : WM == 1 ? (NTAIL ? 3 : 4) : WM == 2 ? 3 : 2) void conv_dw_bf16_kernel(const DwArgs a) {
    static_assert(WM * WN == 4 && (MT == 2 || MT == 4), "four waves, 64 or 128 pixels per wave");
    static_assert(!DEEP || (MT == 2 && !SPLIT), "the two-chunks-ahead form exists for the 64-pixel wave tiles, unsplit");
    static_assert(NTW == 2 || (NTW == 1 && MT == 2), "one n-tile per wave: the 64-pixel wave tile only");
    constexpr int WROWS = 32 * MT;                           // pixels per wave
    constexpr int BM = WROWS * WM, BN = 32 * NTW * WN;
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

    // activation rows: thread -> (16-byte chunk c8 of the 128-byte row segment, rows r0 + 32 i).  a_base: byte offset of the
    // row's top-left tap (may lie "before" the tensor: 32-bit wrap-around arithmetic, only used when the tap is inside the image)
    const int c8 = tid & 7, r0 = tid >> 3;
    // no tap can fall outside the image: scalar tap offsets, no bounds checks.  (DEEP gets the choice as a template argument,
    // PLAINK = 0 / 1: with both paths in its unrolled loop the compiler joins them with a flag and then waits for ALL outstanding
    // loads before it reuses the row registers -- the requests that were meant to stay in flight for two chunks.)
    static_assert(DEEP == (PLAINK >= 0), "the two-chunks-ahead form takes the gather path at compile time");
    const bool plain = PLAINK >= 0 ? PLAINK != 0 : (a.kh == 1 && a.kw == 1 && a.pad == 0);
    unsigned a_base[A_LD];
    int a_ih0[A_LD], a_iw0[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = m < a.M ? m : 0;
        const int t = (int)((unsigned)mm / (unsigned)a.m_w), ow = mm - t * a.m_w;
        const int img = (int)((unsigned)t / (unsigned)a.m_h), oh = t - img * a.m_h;
        a_ih0[i] = m < a.M ? oh * a.stride - a.pad : -0x40000000;    // rows beyond M: no tap is ever inside
        a_iw0[i] = ow * a.stride - a.pad;
        a_base[i] = (unsigned)((((long long)img * a.in_h + a_ih0[i]) * a.in_w + a_iw0[i]) * a.x_ld + a.x_coff + c8 * 8) * 2u;
        if (plain) a_base[i] = m < a.M ? a_base[i] : 0xffffffffu;
    }
    const int st_slot = (c8 ^ ((r0 >> 1) & 7)) * 16;                 // rows r0 + 32 i share (row >> 1) & 7
    char *const st_ptr = smem + r0 * kRowB + st_slot;

    // fragment reads: row = wm WROWS + mt 32 + lr, logical 16-byte chunk 2 s + lh of k-step s
    const int swz = (lr >> 1) & 7;
    int rd_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rd_off[s] = (wm * WROWS + lr) * kRowB + (((2 * s + lh) ^ swz) * 16);

    // weight fragments of this wave's two n-tiles: byte offset = ((nt * ksteps + ks) * 64 + lane) * 16
    // (the packed weights hold an even number of n-tiles; a wave whose 64 channels lie beyond N reads zeros)
    const int wcol = n0 + wn * (32 * NTW);                       // the wave's first channel
    const unsigned w_lane = wcol < a.N ? lane * 16 : 0xffffffffu;
    const bool nt_live0 = !NTAIL || wcol < a.N, nt_live1 = NTW == 2 && (!NTAIL || wcol + 32 < a.N);     // (wave-uniform)
    const int nt0 = wcol >> 5;
    const int w_s0 = __builtin_amdgcn_readfirstlane(nt0 * a.ksteps * kFragB);
    const int w_s1 = __builtin_amdgcn_readfirstlane((nt0 + 1) * a.ksteps * kFragB);

    u32x4 ra[A_LD];
    bf16x8 wf[4][2];
    [[maybe_unused]] u32x4 rb[A_LD];         // DEEP: second set (ra = rows of the odd chunks, rb = of the even ones)
    [[maybe_unused]] bf16x8 wg[4][2];        // DEEP: second set (wf = fragments of the even chunks, wg = of the odd ones)
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    // loader state (uniform): the chunk the next DW_LOAD_A fetches = blocks ld_b, ld_b + 1 of 32 channels, block = (tap (ld_kh,
    // ld_kw), 32-channel block ld_cb of the tap).  Lanes c8 < 4 gather the first block, lanes c8 >= 4 the second (which may
    // belong to the next tap when a tap has an odd number of blocks): tap offset, tap coordinates and "past the end" are
    // per-half values selected per lane.  Past the last chunk every lane's request is out of range and returns zeros without
    // touching memory -- the steady-state loop stays branch-free, which keeps the compiler's vmcnt counts exact.
    // this workgroup's chunks [c_begin, c_end) (all of them without SPLIT)
    int c_begin = 0, c_end = a.nch;
    if constexpr (SPLIT) {
        c_begin = (int)((unsigned)a.nch * blockIdx.y / (unsigned)a.split);
        c_end = (int)((unsigned)a.nch * (blockIdx.y + 1) / (unsigned)a.split);
    }
    const int nblk_end = SPLIT ? min(a.nblk, 2 * c_end) : a.nblk;
    int ld_b = 2 * c_begin, ld_kh = 0, ld_kw = 0, ld_cb = 0;
    if constexpr (SPLIT) {
        const int tap0 = ld_b / a.bpt;
        ld_cb = ld_b - tap0 * a.bpt;
        ld_kh = tap0 / a.kw;
        ld_kw = tap0 - ld_kh * a.kw;
    }
    const bool half1 = c8 >= 4;
#define DW_LOAD_A(R)                                                                                       \
    do {                                                                                                   \
        /* second block of the chunk */                                                                    \
        int kh1_ = ld_kh, kw1_ = ld_kw, cb1_ = ld_cb + 1;                                                  \
        if (cb1_ == a.bpt) { cb1_ = 0; if (++kw1_ == a.kw) { kw1_ = 0; ++kh1_; } }                         \
        const int dy0_ = ld_kh * a.dil, dx0_ = ld_kw * a.dil, dy1_ = kh1_ * a.dil, dx1_ = kw1_ * a.dil;    \
        const bool dead0_ = ld_b >= nblk_end, dead1_ = ld_b + 1 >= nblk_end;                               \
        if (plain) {                                                                                       \
            const bool dead_ = half1 ? dead1_ : dead0_;                                                    \
            const int toff_ = ld_cb * 64;             /* one tap: the two blocks are consecutive channels */ \
            _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                               \
                R[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead_ ? 0xffffffffu : a_base[i], toff_, 0)); \
        } else {                                                                                           \
            /* a_base already holds the lane's channel offset c8 * 8 = 32 (c8 >> 2) + 8 (c8 & 3) */         \
            const unsigned toff0_ = (unsigned)(((dy0_ * a.in_w + dx0_) * a.x_ld + ld_cb * 32) * 2);        \
            const unsigned toff1_ = (unsigned)(((dy1_ * a.in_w + dx1_) * a.x_ld + cb1_ * 32 - 32) * 2);    \
            const unsigned toff_ = half1 ? toff1_ : toff0_;                                                \
            const int dy_ = half1 ? dy1_ : dy0_, dx_ = half1 ? dx1_ : dx0_;                                \
            const bool dead_ = half1 ? dead1_ : dead0_;                                                    \
            _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                             \
                const bool in_ = (unsigned)(a_ih0[i] + dy_) < (unsigned)a.in_h && (unsigned)(a_iw0[i] + dx_) < (unsigned)a.in_w; \
                const unsigned vo_ = (in_ && !dead_) ? a_base[i] + toff_ : 0xffffffffu;                    \
                R[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo_, 0, 0)); \
            }                                                                                              \
        }                                                                                                  \
        /* advance by two blocks */                                                                        \
        ld_b += 2;                                                                                         \
        ld_kh = kh1_; ld_kw = kw1_; ld_cb = cb1_ + 1;                                                      \
        if (ld_cb == a.bpt) { ld_cb = 0; if (++ld_kw == a.kw) { ld_kw = 0; ++ld_kh; } }                    \
    } while (0)
#define DW_STORE_A_(BUF, R)                                                                                \
    do {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                   \
            *reinterpret_cast<u32x4 *>(st_ptr + (BUF) * kBufB + i * 32 * kRowB) = R[i];                    \
    } while (0)
#define DW_STORE_A(BUF) DW_STORE_A_(BUF, ra)
#define DW_LOAD_W_(WF, C, S)                                                                               \
    do {                                                                                                   \
        WF[S][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, w_s0 + ((C) * 4 + (S)) * kFragB, 0)); \
        if constexpr (NTW == 2)                                                                            \
            WF[S][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, w_s1 + ((C) * 4 + (S)) * kFragB, 0)); \
    } while (0)
#define DW_LOAD_W(C, S) DW_LOAD_W_(wf, C, S)
#define DW_READ_A(FA, BUF, S)                                                                              \
    do {                                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
            FA[mt] = *reinterpret_cast<const bf16x8 *>(smem + (BUF) * kBufB + rd_off[S] + mt * 32 * kRowB); \
    } while (0)
// (n-tiles of the wave that lie beyond N -- N = 160 on a 256-wide tile leaves one wave idle and one half idle -- are not
// multiplied: their products would be zeros, but a dense bf16 MFMA loop runs at the chip's power limit, and MFMAs on zeros
// cost the other waves clock)
#define DW_MFMA_(WF, FA, S)                                                                                \
    do {                                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
            if (nt_live0) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[S][0], FA[mt], acc[mt][0], 0, 0, 0); \
            if constexpr (NTW == 2)                                                                        \
                if (nt_live1) acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[S][1], FA[mt], acc[mt][1], 0, 0, 0); \
        }                                                                                                  \
    } while (0)
#define DW_MFMA(FA, S) DW_MFMA_(wf, FA, S)
    constexpr bool kPrefetchA = MT == 4 && WM == 1;      // (the 256 x 128 tile has no registers left for it)
// One k-step.  MT = 4 (two waves per SIMD: the other wave does not always cover an LDS round trip): the fragments of k-step
// S + 1 are requested before the MFMAs of k-step S (second register set); MT = 2 (four waves per SIMD, 128 registers): read and
// multiply in place.
#define DW_MFMA_STEP(BUF, S)                                                                               \
    do {                                                                                                   \
        if constexpr (kPrefetchA) {                                                                        \
            if ((S) == 0) DW_READ_A(fa[0], BUF, 0);                                                        \
            if ((S) < 3) DW_READ_A(fa[((S) + 1) & 1], BUF, ((S) + 1) & 3);                                 \
            DW_SB();                                                                                       \
            DW_MFMA(fa[(S) & 1], S);                                                                       \
        } else {                                                                                           \
            DW_READ_A(fa[0], BUF, S);                                                                      \
            DW_MFMA(fa[0], S);                                                                             \
        }                                                                                                  \
    } while (0)
    bf16x8 fa[kPrefetchA ? 2 : 1][MT];

    // residual rows in the epilogue's layout: lane -> (pixel (lane >> 2) + 16 ps, 8-channel chunk lane & 3) of a 32 x 32 tile;
    // buffer loads with 32-bit offsets (rows beyond M / chunks beyond N: out of range, zeros)
    const int pc = lane & 3, pp = lane >> 2;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, a.res ? (int)a.res_bytes : 0, 0x00020000);
    bf16x8 resq[MT][2][2];
#define DW_FETCH_RES(MTI, NT, PS)                                                                          \
    do {                                                                                                   \
        const int row = m0 + wm * WROWS + (MTI) * 32 + pp + 16 * (PS);                                     \
        const int ch = wcol + (NT) * 32 + 8 * pc;                                                          \
        const unsigned ro = (row < a.M && ch < a.N) ? ((unsigned)row * (unsigned)a.res_ld + ch) * 2u : 0xffffffffu; \
        resq[MTI][NT][PS] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, ro, 0, 0)); \
    } while (0)

    // The order of the memory instructions is pinned with sched_barriers: left alone, the compiler sinks every load of an
    // iteration behind its MFMAs (their destination registers double as LDS fragment registers) and then waits for all of
    // them at the top of the next one -- a full memory latency per chunk.
#define DW_SB() __builtin_amdgcn_sched_barrier(0)
    if constexpr (DEEP) {
        // chunk c: LDS buffer c & 1, fragments in wf (c even) / wg (c odd); its rows were stored one iteration earlier from ra (c odd)
        // / rb (c even).  Requests run two chunks ahead; past the last chunk the row requests are dead (zeros, no memory access) and
        // the fragment requests repeat the last chunk's (the fragment offset is a scalar offset, which the hardware does not check).
        const int last = c_end - c_begin - 1;
#define DW_WCHUNK(C) (c_begin + ((C) < last ? (C) : last))
        {
            u32x4 rp[A_LD];
            DW_LOAD_A(rp);                                  // chunk 0
            DW_LOAD_A(ra);                                  // chunk 1
            DW_SB();
#pragma unroll
            for (int s = 0; s < 4; ++s) DW_LOAD_W_(wf, c_begin, s);
            DW_SB();
            DW_LOAD_A(rb);                                  // chunk 2
            DW_SB();
#pragma unroll
            for (int s = 0; s < 4; ++s) DW_LOAD_W_(wg, DW_WCHUNK(1), s);
            DW_SB();
#pragma unroll
            for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4 *>(st_ptr + i * 32 * kRowB) = rp[i];
        }
        DW_SB();
        __syncthreads();
#define DW_DEEP_ITER(C, BUF, RNEXT, WCUR)                                                                   \
    do {                                                                                                   \
        DW_STORE_A_((BUF) ^ 1, RNEXT);                  /* chunk C + 1, requested two iterations ago */      \
        DW_SB();                                                                                           \
        DW_LOAD_A(RNEXT);                               /* chunk C + 3 */                                    \
        DW_SB();                                                                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
            DW_READ_A(fa[0], BUF, s);                                                                      \
            DW_MFMA_(WCUR, fa[0], s);                                                                      \
            DW_SB();                                                                                       \
            DW_LOAD_W_(WCUR, DW_WCHUNK((C) + 2), s);    /* chunk C + 2 into the registers this k-step has just used */ \
            DW_SB();                                                                                       \
        }                                                                                                  \
        __syncthreads();                                                                                   \
    } while (0)
#define DW_DEEP_LAST(BUF, WCUR)                                                                            \
    do {                                                                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
            DW_READ_A(fa[0], BUF, s);                                                                      \
            DW_MFMA_(WCUR, fa[0], s);                                                                      \
            DW_SB();                                                                                       \
            DW_FETCH_RES(s >> 1, 0, s & 1);                                                                \
            if constexpr (NTW == 2) DW_FETCH_RES(s >> 1, 1, s & 1);                                        \
            DW_SB();                                                                                       \
        }                                                                                                  \
    } while (0)
        int c = 0;
        for (; c + 1 < last; c += 2) {
            DW_DEEP_ITER(c, 0, ra, wf);
            DW_DEEP_ITER(c + 1, 1, rb, wg);
        }
        if (c < last) {                                     // (uniform) one more even chunk before the last, which is odd
            DW_DEEP_ITER(c, 0, ra, wf);
            DW_DEEP_LAST(1, wg);
        } else {
            DW_DEEP_LAST(0, wf);
        }
#undef DW_DEEP_ITER
#undef DW_DEEP_LAST
#undef DW_WCHUNK
    } else {
        // Prologue in the loop's order -- activation rows first, then the fragments -- so that the wait in front of the loop's
        // ds_write counts the same 8 younger fragment loads on the first pass as on every other (chunk 0 in registers of its own).
        {
            u32x4 rp[A_LD];
            DW_LOAD_A(rp);
            DW_LOAD_A(ra);
            DW_SB();
    #pragma unroll
            for (int s = 0; s < 4; ++s) DW_LOAD_W(c_begin, s);
            DW_SB();
    #pragma unroll
            for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4 *>(st_ptr + i * 32 * kRowB) = rp[i];
        }
        DW_SB();
        __syncthreads();
        const int last = c_end - c_begin - 1;
        for (int c = 0; c < last; ++c) {
            const int buf = c & 1;
            DW_STORE_A(buf ^ 1);                            // chunk c + 1, requested one iteration ago (the 8 fragment loads behind it stay in flight)
            DW_SB();
            DW_LOAD_A(ra);                                  // chunk c + 2
            DW_SB();
    #pragma unroll
            for (int s = 0; s < 4; ++s) {
                DW_MFMA_STEP(buf, s);
                DW_SB();
                DW_LOAD_W(c_begin + c + 1, s);              // into the fragment registers this k-step has just used
                DW_SB();
            }
            __syncthreads();
        }
        {
            // last chunk: the residual rows are requested k-step by k-step into the fragment registers it no longer refills
            // (MT = 4: those of the wave's first n-tile; the second n-tile's follow when the epilogue starts)
            const int buf = last & 1;
    #pragma unroll
            for (int s = 0; s < 4; ++s) {
                DW_MFMA_STEP(buf, s);
                DW_SB();
                if constexpr (SPLIT) {
                } else if constexpr (MT == 2) {
                    DW_FETCH_RES(s >> 1, 0, s & 1);
                    if constexpr (NTW == 2) DW_FETCH_RES(s >> 1, 1, s & 1);
                } else {
                    DW_FETCH_RES(s, 0, 0);
                    DW_FETCH_RES(s, 0, 1);
                }
                DW_SB();
            }
        }
    }
    DW_SB();
    __syncthreads();                                    // the epilogue stage reuses the activation buffers
    if constexpr (SPLIT) {
        // raw partial sums, [split][M][N] f32: a lane holds 4 consecutive channels of its pixel per register quad -- 16-byte stores
        float *const ws = a.ws + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int row = m0 + wm * WROWS + mt * 32 + lr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wcol + nt * 32 + 8 * g + 4 * lh;
                    if (row < a.M && ch < a.N) {
                        const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        *reinterpret_cast<f32x4 *>(ws + (size_t)row * a.N + ch) = v;
                    }
                }
            }
        return;
    }
    if constexpr (MT == 4) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            DW_FETCH_RES(mt, 1, 0);
            DW_FETCH_RES(mt, 1, 1);
        }
        DW_SB();
    }

    // epilogue.  acc[mt][nt][4 g + i] = C[pixel m0 + wm WROWS + 32 mt + lr][channel n0 + wn 64 + 32 nt + 8 g + 4 lh + i].
    // The loop ended on a barrier: the activation buffers are dead, each wave stages its tiles in its own 32 x 36 f32 slice.
    float *const stage = reinterpret_cast<float *>(smem) + wave * (32 * kStageLd);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
    const int prow0 = m0 + wm * WROWS, pcol0 = wcol;
    // ReLU on the packed bf16 pairs after the rounding (rounding keeps sign and order, so relu(round(v)) == round(relu(v))):
    // a signed 16-bit max with 0 clears every negative value; without ReLU the floor is the most negative pattern (identity).
    // 4 instructions per 8 values instead of the 16 of fmaxf on f32 (canonicalise + max).
    const unsigned floor2 = a.relu ? 0u : 0x80008000u;
    // output row offsets of the wave's pixels (bytes, 32-bit).  Transposed convolution: GEMM row = input pixel (img, ih, iw) ->
    // output pixel (ih ks, iw ks) of the same image; the tap part of the address comes with the column
    unsigned yrow[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = prow0 + mt * 32 + pp + 16 * ps;
            unsigned pix = (unsigned)row;
            if (a.ks > 0) {
                const int rr = row < a.M ? row : 0;
                const int t = (int)((unsigned)rr / (unsigned)a.in_w), iw = rr - t * a.in_w;
                const int img = (int)((unsigned)t / (unsigned)a.in_h), ih = t - img * a.in_h;
                pix = (unsigned)((img * a.out_h + ih * a.ks) * a.out_w + iw * a.ks);
            }
            yrow[mt][ps] = row < a.M ? pix * (unsigned)(a.y_ld * 2) : 0xffffffffu;
        }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int ch = pcol0 + nt * 32 + 8 * pc;
        const bool ch_ok = ch < a.N;
        int co = ch;                                                       // output channel of the 8-channel chunk
        unsigned ycol = (unsigned)(a.y_coff + ch) * 2u;
        if (a.ks > 0) {                                                    // (cout % 8 == 0: a chunk stays inside one tap)
            const int tap = (int)((unsigned)ch / (unsigned)a.cout);
            co = ch - tap * a.cout;
            const int dy = (int)((unsigned)tap / (unsigned)a.ks), dx = tap - dy * a.ks;
            ycol = (unsigned)((dy * a.out_w + dx) * a.y_ld + a.y_coff + co) * 2u;
        }
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 sc0 = one, sc1 = one, sh0 = zero, sh1 = zero;
        if (ch_ok && a.scale) { sc0 = *reinterpret_cast<const f32x4 *>(a.scale + co); sc1 = *reinterpret_cast<const f32x4 *>(a.scale + co + 4); }
        if (ch_ok && a.bias) { sh0 = *reinterpret_cast<const f32x4 *>(a.bias + co); sh1 = *reinterpret_cast<const f32x4 *>(a.bias + co + 4); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stage + lr * kStageLd + 8 * g + 4 * lh) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int p = pp + 16 * ps;
                f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc);
                f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc + 4);
                v0 = v0 * sc0 + sh0;
                v1 = v1 * sc1 + sh1;
                const bf16x8 rq = resq[mt][nt][ps];                       // (zeros when there is no residual)
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] += (float)rq[i]; v1[i] += (float)rq[4 + i]; }
                const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
                u32x4 o = __builtin_bit_cast(u32x4, __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned d_;
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(d_) : "v"(o[i]), "v"(floor2));
                    o[i] = d_;
                }
                // rows beyond M / chunks beyond N: out-of-range offset, the store is dropped (no branch)
                const unsigned yo = (yrow[mt][ps] != 0xffffffffu && ch_ok) ? yrow[mt][ps] + ycol : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128(o, y_rsrc, yo, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// TWO LAYERS IN ONE KERNEL: a k x k convolution with 256 outputs (folded BN, ReLU) followed by a 1x1 convolution over those 256
// channels with residual and ReLU -- conv2 + conv3 of a ResNet layer-3 bottleneck (23 of them in ResNet-101).  Phase 1 is the
// 64 x 256 tile of the kernel above; instead of storing its tile it rounds it to bf16 into LDS in the layout the fragment reads
// expect (the wave that owns channels 64 w .. 64 w + 63 writes k-chunk w of the second GEMM: 64 pixels x 128 B, same swizzle),
// and after ONE barrier every wave runs the second GEMM on its own: K2 = 256 from that tile, N2 in chunks of 256 columns
// (64 per wave), the second layer's weights again streamed from L2 in fragment order, the residual rows of a chunk requested
// during its last k-chunk, the common epilogue per chunk.  The 256-channel map between the two layers never exists in HBM
// (17 + 17 MB per layer at cfg-3), the second layer has no activation loads, and one launch replaces two.
// Same arithmetic as the two launches (f32 accumulation in the same k order, the middle map rounded to bf16 once): bitwise the
// result of sgv3d_conv_dw_bf16_forward twice.
__global__ __launch_bounds__(256, 3) void conv_dw_bf16_pair_kernel(const DwArgs a) {
    constexpr int WM = 1, MT = 2, NTW = 2;                    // (four waves side by side along the 256 channels, two 32-channel tiles each)
    constexpr int WROWS = 32 * MT, BM = WROWS * WM;
    constexpr int A_LD = BM / 32;
    constexpr int kBufB = BM * kRowB;
    constexpr int kRegion0 = 4 * 32 * kStageLd * 4;          // 18 432 B: activation buffers (2 x 8 KB) in phase 1, epilogue stage in phase 2
    constexpr bool kPrefetchA = false;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const mid = smem + kRegion0;                         // [4 k-chunks][64 pixels][128 B]: the first layer's tile as bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    [[maybe_unused]] const int wm = 0;
    const int wn = __builtin_amdgcn_readfirstlane(wave);
    const int lr = lane & 31, lh = lane >> 5;
    const int ntiles = a.tiles_m;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    [[maybe_unused]] const int n0 = 0;
    const int m0 = tm * BM;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.w_bytes, 0x00020000);
    const int c8 = tid & 7, r0 = tid >> 3;
    const bool plain = a.kh == 1 && a.kw == 1 && a.pad == 0;
    unsigned a_base[A_LD];
    int a_ih0[A_LD], a_iw0[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = m < a.M ? m : 0;
        const int t = (int)((unsigned)mm / (unsigned)a.m_w), ow = mm - t * a.m_w;
        const int img = (int)((unsigned)t / (unsigned)a.m_h), oh = t - img * a.m_h;
        a_ih0[i] = m < a.M ? oh * a.stride - a.pad : -0x40000000;
        a_iw0[i] = ow * a.stride - a.pad;
        a_base[i] = (unsigned)((((long long)img * a.in_h + a_ih0[i]) * a.in_w + a_iw0[i]) * a.x_ld + a.x_coff + c8 * 8) * 2u;
        if (plain) a_base[i] = m < a.M ? a_base[i] : 0xffffffffu;
    }
    const int st_slot = (c8 ^ ((r0 >> 1) & 7)) * 16;
    char *const st_ptr = smem + r0 * kRowB + st_slot;
    const int swz = (lr >> 1) & 7;
    int rd_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rd_off[s] = lr * kRowB + (((2 * s + lh) ^ swz) * 16);
    const unsigned w_lane = lane * 16;
    constexpr bool nt_live0 = true, nt_live1 = true;
    const int nt0 = wn * 2;
    const int w_s0 = __builtin_amdgcn_readfirstlane(nt0 * a.ksteps * kFragB);
    const int w_s1 = __builtin_amdgcn_readfirstlane((nt0 + 1) * a.ksteps * kFragB);
    u32x4 ra[A_LD];
    bf16x8 wf[4][2];
    bf16x8 fa[1][MT];
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    const int nblk_end = a.nblk;
    int ld_b = 0, ld_kh = 0, ld_kw = 0, ld_cb = 0;
    const bool half1 = c8 >= 4;
    // ---- phase 1: the k x k layer (the loop of conv_dw_bf16_kernel<1, 4, 2>) -------------------------------------------------
    {
        u32x4 rp[A_LD];
        DW_LOAD_A(rp);
        DW_LOAD_A(ra);
        DW_SB();
#pragma unroll
        for (int s = 0; s < 4; ++s) DW_LOAD_W(0, s);
        DW_SB();
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4 *>(st_ptr + i * 32 * kRowB) = rp[i];
    }
    DW_SB();
    __syncthreads();
    const int last = a.nch - 1;
    for (int c = 0; c < last; ++c) {
        const int buf = c & 1;
        DW_STORE_A(buf ^ 1);
        DW_SB();
        DW_LOAD_A(ra);
        DW_SB();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            DW_MFMA_STEP(buf, s);
            DW_SB();
            DW_LOAD_W(c + 1, s);
            DW_SB();
        }
        __syncthreads();
    }
    // second layer's fragments: n-tiles (n2 * 8 + wn * 2, + 1), k-steps c * 4 + s of ksteps2 = 16
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, (int)a.w2_bytes, 0x00020000);
    const int nch2 = a.N2 >> 8;                              // chunks of 256 output columns
#define PAIR_LOAD_W2(N2I, C, S, LIVE)                                                                     \
    do {                                                                                                   \
        const int b0_ = (((N2I) * 8 + nt0) * a.ksteps2 + (C) * 4 + (S)) * kFragB;                          \
        const unsigned vl_ = (LIVE) ? w_lane : 0xffffffffu;                                                \
        wf[S][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2_rsrc, vl_, b0_, 0)); \
        wf[S][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2_rsrc, vl_, b0_ + a.ksteps2 * kFragB, 0)); \
    } while (0)
    {
        const int buf = last & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            DW_MFMA_STEP(buf, s);
            DW_SB();
            PAIR_LOAD_W2(0, 0, s, true);                    // the second GEMM's first fragments, under the first one's tail
            DW_SB();
        }
    }
    // ---- the first layer's tile -> LDS as bf16 (folded BN, ReLU), k-chunk wn of the second GEMM --------------------------------
    {
        const unsigned floor1 = a.relu ? 0u : 0x80008000u;
        char *const mw = mid + wn * (64 * kRowB);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wn * 64 + nt * 32 + 8 * g + 4 * lh;
                const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + ch) : one;
                const f32x4 sh = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + ch) : zero;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    v = v * sc + sh;
                    union { bf16x4 b; unsigned u[2]; } pk;
                    pk.b = __builtin_convertvector(v, bf16x4);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        unsigned d_;
                        asm("v_pk_max_i16 %0, %1, %2" : "=v"(d_) : "v"(pk.u[i]), "v"(floor1));
                        pk.u[i] = d_;
                    }
                    const int row = mt * 32 + lr;
                    *reinterpret_cast<unsigned long long *>(mw + row * kRowB + (((nt * 4 + g) ^ swz) * 16) + lh * 8) =
                        (unsigned long long)pk.u[0] | ((unsigned long long)pk.u[1] << 32);
                }
            }
    }
    __syncthreads();               // the tile is complete; the activation buffers (= the epilogue stage) are dead
    // ---- phase 2: the 1x1 layer, every wave on its own ------------------------------------------------------------------------
    const int pc = lane & 3, pp = lane >> 2;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, a.res ? (int)a.res_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
    float *const stage = reinterpret_cast<float *>(smem) + wave * (32 * kStageLd);
    const unsigned floor2 = a.relu2 ? 0u : 0x80008000u;
    const int prow0 = m0;
    unsigned yrow[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = prow0 + mt * 32 + pp + 16 * ps;
            yrow[mt][ps] = row < a.M ? (unsigned)row * (unsigned)(a.y_ld * 2) : 0xffffffffu;
        }
    for (int n2 = 0; n2 < nch2; ++n2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        const int pcol0 = n2 * 256 + wn * 64;
        bf16x8 resq[MT][2][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[0][mt] = *reinterpret_cast<const bf16x8 *>(mid + c * (64 * kRowB) + rd_off[s] + mt * 32 * kRowB);
                DW_MFMA(fa[0], s);
                DW_SB();
                if (c < 3) PAIR_LOAD_W2(n2, c + 1, s, true);
                else PAIR_LOAD_W2(n2 + 1, 0, s, n2 + 1 < nch2);
                if (c == 3) {                                  // the residual rows of this chunk, under its last k-chunk
                    const int row = prow0 + (s >> 1) * 32 + pp + 16 * (s & 1);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int ch = pcol0 + nt * 32 + 8 * pc;
                        const unsigned ro = (row < a.M && ch < a.N2) ? ((unsigned)row * (unsigned)a.res_ld + ch) * 2u : 0xffffffffu;
                        resq[s >> 1][nt][s & 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, ro, 0, 0));
                    }
                }
                DW_SB();
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ch = pcol0 + nt * 32 + 8 * pc;
            const bool ch_ok = ch < a.N2;
            const unsigned ycol = (unsigned)(a.y_coff + ch) * 2u;
            const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 sc0 = one, sc1 = one, sh0 = zero, sh1 = zero;
            if (ch_ok && a.scale2) { sc0 = *reinterpret_cast<const f32x4 *>(a.scale2 + ch); sc1 = *reinterpret_cast<const f32x4 *>(a.scale2 + ch + 4); }
            if (ch_ok && a.bias2) { sh0 = *reinterpret_cast<const f32x4 *>(a.bias2 + ch); sh1 = *reinterpret_cast<const f32x4 *>(a.bias2 + ch + 4); }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(stage + lr * kStageLd + 8 * g + 4 * lh) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int p = pp + 16 * ps;
                    f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc);
                    f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + p * kStageLd + 8 * pc + 4);
                    v0 = v0 * sc0 + sh0;
                    v1 = v1 * sc1 + sh1;
                    const bf16x8 rq = resq[mt][nt][ps];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v0[i] += (float)rq[i]; v1[i] += (float)rq[4 + i]; }
                    const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
                    u32x4 o = __builtin_bit_cast(u32x4, __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        unsigned d_;
                        asm("v_pk_max_i16 %0, %1, %2" : "=v"(d_) : "v"(o[i]), "v"(floor2));
                        o[i] = d_;
                    }
                    const unsigned yo = (yrow[mt][ps] != 0xffffffffu && ch_ok) ? yrow[mt][ps] + ycol : 0xffffffffu;
                    __builtin_amdgcn_raw_buffer_store_b128(o, y_rsrc, yo, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#undef PAIR_LOAD_W2
}

#undef DW_SB
#undef DW_LOAD_A
#undef DW_STORE_A
#undef DW_STORE_A_
#undef DW_LOAD_W_
#undef DW_MFMA_
#undef DW_LOAD_W
#undef DW_FETCH_RES
#undef DW_MFMA_STEP
#undef DW_MFMA
#undef DW_READ_A

// split-K epilogue: sums the partial tiles in split order, then the epilogue of the main kernel (scale / shift, residual, ReLU on the
// rounded value, bf16) -- one thread per 8 channels of a pixel
__global__ __launch_bounds__(256) void dw_splitk_reduce_kernel(const DwArgs a) {
    const int nq = a.N >> 3;
    const long long total = (long long)a.M * nq;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int row = (int)(i / nq), ch = (int)(i - (long long)row * nq) * 8;
    const float *p = a.ws + (size_t)row * a.N + ch;
    const size_t plane = (size_t)a.M * a.N;
    f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + 4);
    for (int s = 1; s < a.split; ++s) {
        v0 += *reinterpret_cast<const f32x4 *>(p + s * plane);
        v1 += *reinterpret_cast<const f32x4 *>(p + s * plane + 4);
    }
    if (a.scale) { v0 = v0 * *reinterpret_cast<const f32x4 *>(a.scale + ch); v1 = v1 * *reinterpret_cast<const f32x4 *>(a.scale + ch + 4); }
    if (a.bias) { v0 += *reinterpret_cast<const f32x4 *>(a.bias + ch); v1 += *reinterpret_cast<const f32x4 *>(a.bias + ch + 4); }
    if (a.res) {
        const bf16x8 rq = *reinterpret_cast<const bf16x8 *>(static_cast<const __bf16 *>(a.res) + (size_t)row * a.res_ld + ch);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v0[k] += (float)rq[k]; v1[k] += (float)rq[4 + k]; }
    }
    const bf16x4 o0 = __builtin_convertvector(v0, bf16x4), o1 = __builtin_convertvector(v1, bf16x4);
    u32x4 o = __builtin_bit_cast(u32x4, __builtin_shufflevector(o0, o1, 0, 1, 2, 3, 4, 5, 6, 7));
    const unsigned floor2 = a.relu ? 0u : 0x80008000u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned d_;
        asm("v_pk_max_i16 %0, %1, %2" : "=v"(d_) : "v"(o[k]), "v"(floor2));
        o[k] = d_;
    }
    *reinterpret_cast<u32x4 *>(static_cast<__bf16 *>(a.y) + (size_t)row * a.y_ld + a.y_coff + ch) = o;
}

template <int WM, int WN, int MT, bool NTAIL, bool SPLIT, bool DEEP = false, int NTW = 2, int PLAINK = -1>
int launch_dw_t(const DwArgs &a0, hipStream_t st) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NTW * WN;
    DwArgs a = a0;
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.N, BN);
    constexpr size_t tiles = 2 * (size_t)BM * kRowB, stage = sizeof(float) * 4 * 32 * kStageLd;
    constexpr size_t lds = tiles > stage ? tiles : stage;
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_dw_bf16_kernel<WM, WN, MT, NTAIL, SPLIT, DEEP, NTW, PLAINK>), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv_dw_bf16: cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL((conv_dw_bf16_kernel<WM, WN, MT, NTAIL, SPLIT, DEEP, NTW, PLAINK>), dim3(a.tiles_m * a.tiles_n, SPLIT ? a.split : 1), dim3(256), lds, st, a);
    if constexpr (SPLIT) {
        if (int rc = check_launch("conv_dw_bf16_kernel")) return rc;
        hipLaunchKernelGGL(dw_splitk_reduce_kernel, dim3((unsigned)cdiv((long long)a.M * (a.N >> 3), 256)), dim3(256), 0, st, a);
        return check_launch("dw_splitk_reduce_kernel");
    }
    return check_launch("conv_dw_bf16_kernel");
}

template <int WM, int WN>
int launch_dw_deep(const DwArgs &a, hipStream_t st) {
    if (a.split > 1) return fail(SGV3D_EINVAL, "conv_dw_bf16: the *_DEEP tiles do not split along k");
    const bool plain = a.kh == 1 && a.kw == 1 && a.pad == 0;
    if (a.N % (64 * WN) == 0) return plain ? launch_dw_t<WM, WN, 2, false, false, true, 2, 1>(a, st) : launch_dw_t<WM, WN, 2, false, false, true, 2, 0>(a, st);
    return plain ? launch_dw_t<WM, WN, 2, true, false, true, 2, 1>(a, st) : launch_dw_t<WM, WN, 2, true, false, true, 2, 0>(a, st);
}

// 64 pixels x 128 channels: one 32-channel tile per wave
int launch_dw_narrow(const DwArgs &a, hipStream_t st) {
    if (a.split > 1)
        return a.N % 128 == 0 ? launch_dw_t<1, 4, 2, false, true, false, 1>(a, st) : launch_dw_t<1, 4, 2, true, true, false, 1>(a, st);
    return a.N % 128 == 0 ? launch_dw_t<1, 4, 2, false, false, false, 1>(a, st) : launch_dw_t<1, 4, 2, true, false, false, 1>(a, st);
}

int launch_dw_narrow_deep(const DwArgs &a, hipStream_t st) {
    if (a.split > 1) return fail(SGV3D_EINVAL, "conv_dw_bf16: the *_DEEP tiles do not split along k");
    const bool plain = a.kh == 1 && a.kw == 1 && a.pad == 0;
    if (a.N % 128 == 0) return plain ? launch_dw_t<1, 4, 2, false, false, true, 1, 1>(a, st) : launch_dw_t<1, 4, 2, false, false, true, 1, 0>(a, st);
    return plain ? launch_dw_t<1, 4, 2, true, false, true, 1, 1>(a, st) : launch_dw_t<1, 4, 2, true, false, true, 1, 0>(a, st);
}

template <int WM, int WN, int MT>
int launch_dw(const DwArgs &a, hipStream_t st) {
    if (a.split > 1)
        return a.N % (64 * WN) == 0 ? launch_dw_t<WM, WN, MT, false, true>(a, st) : launch_dw_t<WM, WN, MT, true, true>(a, st);
    return a.N % (64 * WN) == 0 ? launch_dw_t<WM, WN, MT, false, false>(a, st) : launch_dw_t<WM, WN, MT, true, false>(a, st);
}

}  // namespace

extern "C" size_t sgv3d_conv_dw_bf16_weight_bytes(int cout, int cin, int kh, int kw) {
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return 0;
    return (size_t)cdiv(cout, 64) * 2 * ((size_t)cdiv((long long)kh * kw * cdiv(cin, 32), 2) * 4) * kFragB;
}

extern "C" int sgv3d_conv_dw_bf16_pack_weight(const float *w, int cout, int cin_w, int cin, int kh, int kw, void *w_packed, void *stream) {
    SGV3D_REQUIRE(w && w_packed && cout > 0 && cin_w > 0 && cin >= cin_w && kh > 0 && kw > 0, "conv_dw_bf16_pack_weight: bad argument");
    const int bpt = cdiv(cin, 32), ksteps = cdiv((long long)kh * kw * bpt, 2) * 4;
    hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv(cout, 64) * 2 * ksteps), dim3(64), 0, as_stream(stream), w, cout, cin_w, kh * kw, bpt, ksteps,
                       static_cast<__bf16 *>(w_packed));
    return check_launch("dw_pack_kernel");
}

// f32 partial tiles of a split-K launch: [split_k][M][cout]
extern "C" size_t sgv3d_conv_dw_bf16_workspace_bytes(const sgv3d_conv_desc *d) {
    if (!d || d->split_k <= 1 || d->batch <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->cout <= 0) return 0;
    return (size_t)d->split_k * d->batch * d->out_h * d->out_w * d->cout * sizeof(float);
}

extern "C" int sgv3d_conv_dw_bf16_forward_splitk(const sgv3d_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                                 const float *bias, const void *residual, void *y, void *workspace,
                                                 size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(d && x && w_packed && y, "conv_dw_bf16: null pointer");
    const bool deconv = d->mode == SGV3D_CONV_DECONV;
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL || deconv, "conv_dw_bf16: NORMAL / DECONV mode only");
    const int split = d->split_k > 1 ? d->split_k : 1;
    if (split > 1) {
        SGV3D_REQUIRE(!deconv, "conv_dw_bf16: split-K covers NORMAL mode only");
        SGV3D_REQUIRE(d->kh > 0 && d->kw > 0 && d->cin > 0 && split <= cdiv(d->kh * d->kw * (d->cin / 32), 2),
                      "conv_dw_bf16: split_k %d exceeds the layer's k chunks of 64", split);
        const size_t need = sgv3d_conv_dw_bf16_workspace_bytes(d);
        if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
            return fail(SGV3D_ENOSPACE, "conv_dw_bf16: split-K needs %zu B of 16-B aligned workspace (got %zu)", need, workspace_bytes);
    }
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 &&
                      d->dil > 0 && d->pad >= 0, "conv_dw_bf16: non-positive dimension");
    if (deconv) {
        SGV3D_REQUIRE(d->deconv_ks >= 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->dil == 1 && residual == nullptr,
                      "conv_dw_bf16: DECONV runs as a 1x1 GEMM with deconv_ks = kernel = stride, no residual");
        SGV3D_REQUIRE(d->out_h == d->in_h * d->deconv_ks && d->out_w == d->in_w * d->deconv_ks, "conv_dw_bf16: DECONV output must be input * deconv_ks");
    } else {
        const int eh = (d->in_h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
        const int ew = (d->in_w + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
        SGV3D_REQUIRE(eh == d->out_h && ew == d->out_w, "conv_dw_bf16: output %dx%d does not match the conv arithmetic %dx%d", d->out_h, d->out_w, eh, ew);
    }
    SGV3D_REQUIRE(d->cin % 32 == 0 && d->cout % 8 == 0, "conv_dw_bf16: cin must be a multiple of 32 and cout of 8 (got %d / %d)", d->cin, d->cout);
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout && (residual == nullptr || d->res_ld >= d->cout),
                  "conv_dw_bf16: channel strides too small");
    SGV3D_REQUIRE(d->x_ld % 8 == 0 && d->x_coff % 8 == 0 && d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && (residual == nullptr || d->res_ld % 8 == 0),
                  "conv_dw_bf16: channel strides / offsets must be multiples of 8 (16-byte rows of bf16)");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                  "conv_dw_bf16: pointers must be 16-B aligned");
    const int gemm_n = deconv ? d->cout * d->deconv_ks * d->deconv_ks : d->cout;
    const long long M = (long long)d->batch * (deconv ? (long long)d->in_h * d->in_w : (long long)d->out_h * d->out_w);
    const long long xb = (long long)d->batch * d->in_h * d->in_w * d->x_ld * 2;
    const long long yb = (long long)d->batch * d->out_h * d->out_w * d->y_ld * 2;
    const size_t wb = sgv3d_conv_dw_bf16_weight_bytes(gemm_n, d->cin, d->kh, d->kw);
    SGV3D_REQUIRE(M < 0x7fffffffLL && xb < 0xf0000000LL && wb < 0xf0000000ULL && (residual == nullptr || M * d->res_ld * 2 < 0xf0000000LL) &&
                      yb < 0xf0000000LL,
                  "conv_dw_bf16: operands larger than 3.75 GiB (32-bit buffer offsets)");
    SGV3D_REQUIRE((long long)d->dil * (d->kh - 1) < 0x10000 && d->in_h < 0x10000000 && d->in_w < 0x10000000, "conv_dw_bf16: kernel extent too large");
    DwArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.M = (int)M; a.N = gemm_n; a.cin = d->cin;
    a.ks = deconv ? d->deconv_ks : 0; a.cout = d->cout;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld; a.relu = d->relu;
    a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w;
    a.m_h = deconv ? d->in_h : d->out_h; a.m_w = deconv ? d->in_w : d->out_w;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.bpt = d->cin / 32;
    a.nblk = d->kh * d->kw * a.bpt;
    a.nch = cdiv(a.nblk, 2);
    a.ksteps = a.nch * 4;
    a.tiles_m = a.tiles_n = 0;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.res_bytes = residual ? (unsigned)(M * d->res_ld * 2) : 0u;
    a.y_bytes = (unsigned)yb;
    a.split = split; a.ws = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    switch (d->tile) {
        case SGV3D_TILE_DW_64x256: return launch_dw<1, 4, 2>(a, st);
        case SGV3D_TILE_DW_128x128: return launch_dw<2, 2, 2>(a, st);
        case SGV3D_TILE_DW_256x64: return launch_dw<4, 1, 2>(a, st);
        case SGV3D_TILE_DW_128x256: return launch_dw<1, 4, 4>(a, st);
        case SGV3D_TILE_DW_256x128: return launch_dw<2, 2, 4>(a, st);
        case SGV3D_TILE_DW_64x256_DEEP: return launch_dw_deep<1, 4>(a, st);
        case SGV3D_TILE_DW_128x128_DEEP: return launch_dw_deep<2, 2>(a, st);
        case SGV3D_TILE_DW_64x128: return launch_dw_narrow(a, st);
        case SGV3D_TILE_DW_64x128_DEEP: return launch_dw_narrow_deep(a, st);
        default: return fail(SGV3D_EINVAL, "conv_dw_bf16: desc.tile must be one of SGV3D_TILE_DW_* (got %d)", d->tile);
    }
}

extern "C" int sgv3d_conv_dw_bf16_forward(const sgv3d_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                          const float *bias, const void *residual, void *y, void *stream) {
    SGV3D_REQUIRE(d && d->split_k <= 1, "conv_dw_bf16: split-K takes a workspace (sgv3d_conv_dw_bf16_forward_splitk)");
    return sgv3d_conv_dw_bf16_forward_splitk(d, x, w_packed, scale, bias, residual, y, nullptr, 0, stream);
}

// conv A (k x k, 256 outputs, folded BN + ReLU as desc says) followed by conv B (1x1 over those 256 channels, cout2 outputs,
// folded BN, residual, ReLU) in one launch (conv_dw_bf16_pair_kernel).
extern "C" int sgv3d_conv_dw_bf16_pair_forward(const sgv3d_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                               const float *bias, int cout2, const void *w2_packed, const float *scale2,
                                               const float *bias2, const void *residual, int res_ld, void *y, int y_ld, int y_coff,
                                               int relu2, void *stream) {
    SGV3D_REQUIRE(d && x && w_packed && w2_packed && y, "conv_dw_bf16_pair: null pointer");
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL && d->split_k <= 1, "conv_dw_bf16_pair: NORMAL mode without split-K only");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0,
                  "conv_dw_bf16_pair: non-positive dimension");
    SGV3D_REQUIRE(d->cout == 256 && cout2 > 0 && cout2 % 256 == 0, "conv_dw_bf16_pair: the first layer must have 256 outputs, the second a multiple of 256 (got %d / %d)",
                  d->cout, cout2);
    const int eh = (d->in_h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int ew = (d->in_w + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    SGV3D_REQUIRE(eh == d->out_h && ew == d->out_w, "conv_dw_bf16_pair: output %dx%d does not match the conv arithmetic %dx%d", d->out_h, d->out_w, eh, ew);
    SGV3D_REQUIRE(d->cin % 32 == 0 && d->x_ld >= d->x_coff + d->cin && y_ld >= y_coff + cout2 && (residual == nullptr || res_ld >= cout2),
                  "conv_dw_bf16_pair: cin %% 32, channel strides");
    SGV3D_REQUIRE(d->x_ld % 8 == 0 && d->x_coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0 && (residual == nullptr || res_ld % 8 == 0),
                  "conv_dw_bf16_pair: channel strides / offsets must be multiples of 8 (16-byte rows of bf16)");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(w2_packed) | reinterpret_cast<uintptr_t>(scale) |
                    reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(scale2) | reinterpret_cast<uintptr_t>(bias2)) & 15) == 0,
                  "conv_dw_bf16_pair: pointers must be 16-B aligned");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    const long long xb = (long long)d->batch * d->in_h * d->in_w * d->x_ld * 2, yb = M * y_ld * 2;
    const size_t wb = sgv3d_conv_dw_bf16_weight_bytes(256, d->cin, d->kh, d->kw), wb2 = sgv3d_conv_dw_bf16_weight_bytes(cout2, 256, 1, 1);
    SGV3D_REQUIRE(M < 0x7fffffffLL && xb < 0xf0000000LL && yb < 0xf0000000LL && wb < 0xf0000000ULL && wb2 < 0xf0000000ULL &&
                      (residual == nullptr || M * res_ld * 2 < 0xf0000000LL),
                  "conv_dw_bf16_pair: operands larger than 3.75 GiB (32-bit buffer offsets)");
    SGV3D_REQUIRE((long long)d->dil * (d->kh - 1) < 0x10000 && d->in_h < 0x10000000 && d->in_w < 0x10000000, "conv_dw_bf16_pair: kernel extent too large");
    DwArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.M = (int)M; a.N = 256; a.cin = d->cin;
    a.ks = 0; a.cout = 256;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = y_ld; a.y_coff = y_coff; a.res_ld = res_ld; a.relu = d->relu;
    a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w;
    a.m_h = d->out_h; a.m_w = d->out_w;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.bpt = d->cin / 32;
    a.nblk = d->kh * d->kw * a.bpt;
    a.nch = cdiv(a.nblk, 2);
    a.ksteps = a.nch * 4;
    a.tiles_m = cdiv(M, 64); a.tiles_n = 1;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.res_bytes = residual ? (unsigned)(M * res_ld * 2) : 0u; a.y_bytes = (unsigned)yb;
    a.w2 = w2_packed; a.scale2 = scale2; a.bias2 = bias2; a.N2 = cout2; a.relu2 = relu2; a.ksteps2 = 16; a.w2_bytes = (unsigned)wb2;
    constexpr size_t lds = 4 * 32 * kStageLd * 4 + 4 * 64 * kRowB;
    static PerDeviceSize lds_set;
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_dw_bf16_pair_kernel), lds, lds_set))
        return fail(SGV3D_ELAUNCH, "conv_dw_bf16_pair: cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL(conv_dw_bf16_pair_kernel, dim3(a.tiles_m), dim3(256), lds, as_stream(stream), a);
    return check_launch("conv_dw_bf16_pair_kernel");
}
