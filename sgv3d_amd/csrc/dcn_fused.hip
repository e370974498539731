// Deformable 3x3 convolution (mmcv 1.4.0 DeformConv2dPack / DCNv1 as configured at layers/backbones/lss_fpn.py:190-198: 3x3,
// stride 1, pad 1, dilation 1, groups 4, deform_groups 1, no bias) as ONE implicit GEMM whose A operand is sampled on the fly:
//
//   out[p][g * opg + co] = sum_tap sum_ci W[g][co][tap * cpg + ci] * bilinear(x[:, :, g * cpg + ci], p + tap + offset[p][tap])
//
// The round-1..4 form materialised the sampled column tensor (deform_im2col3x3: 95.5 MB written, 181 MB of HBM traffic per
// cfg-2 launch, profiles/r04_hbm_traffic.json) and ran four grouped GEMMs that re-read it.  Here the bilinear samples go
// straight from the input map into the GEMM's LDS stage: no column tensor, one launch for all groups.
//
//   workgroup   256 threads = 4 waves; tile 64 pixels x 64 output channels (of one group) x 32 k; the GEMM core of
//               gemm16_grouped.hip (v_mfma_f32_16x16x4_f32, weights on M so that a lane ends up with 4 consecutive output
//               channels of a pixel, k order k = 16 q + 4 (lane / 16) + j in both operands, XOR-swizzled 128-byte LDS rows,
//               one register stage, one barrier per k-tile).  A wave owns all 64 pixels x 16 channels.
//   A operand   k-tile kt = (tap, 32-channel chunk of the group).  Thread (row r0 = tid / 8 [+32], chunk cc = tid % 8) owns 4
//               channels of two pixels: per tap it derives the four corner offsets (out of range for corners outside the
//               image: the buffer load returns zeros, as the reference's zero padding) and the four bilinear weights of its two
//               pixels from the (dy, dx) pairs staged in LDS once per workgroup; per k-tile it issues 8 16-byte loads and
//               combines them exactly as deform_im2col3x3_kernel did (w1 v1 + w2 v2 + w3 v3 + w4 v4, same order).
//   epilogue    raw accumulators to out[pixel][y_coff + g * opg + co] (the layer has neither bias nor norm).
//
// Bound: MFMA f32 (157.3 TFLOP/s): 2 x B H W x 9 cpg x cout flop per launch; HBM: the input map and the offsets once
// (L2 serves the nine overlapping taps), the output once.  The sampling arithmetic is vector-ALU work that shares the SIMD
// with the f32 MFMAs (~50 instructions per thread and k-tile beside 32 MFMAs of 32 cycles).
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;
constexpr int kThreads = 256;
constexpr int kMaxGroups = 8;

struct DcnArgs {
    const float *x, *off;
    const float *w[kMaxGroups];     // per group: packed [cout_pad][k_pad], k = tap * cpg + ci
    float *y;
    int H, W, C, cpg, opg, groups;
    int M;                          // B * H * W
    int off_ld, y_ld, y_coff, k_pad;
    int tiles_m, tiles_ng;          // tiles_ng = groups * ceil(opg / 64)
    unsigned x_bytes, w_bytes;
};

__global__ __launch_bounds__(kThreads, 3) void dcn3x3_fused_kernel(const DcnArgs a) {
    constexpr int MB = 4, BM = 64, BN = 64;
    constexpr int kBuf = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) float smem[2 * kBuf];
    __shared__ float off_s[BM][18];
    float *const Xs0 = smem;
    float *const Ws0 = smem + BM * BK;

    const int ntiles = a.tiles_m * a.tiles_ng;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tng = (int)((unsigned)logical / (unsigned)a.tiles_m);      // (group, n-tile): changes slowest -> shared weight panel
    const int tm = logical - tng * a.tiles_m;
    const int tn_per_g = a.tiles_ng / a.groups;
    const int grp = tng / tn_per_g;
    const int n0 = (tng - grp * tn_per_g) * BN;                          // first output channel inside the group
    const int m0 = tm * BM;

    const int tid = threadIdx.x;
    const int cc = tid & 7, r0 = tid >> 3;
    const int cs = cc ^ ((r0 >> 1) & 7);

    // ---- offsets of the tile's 64 pixels -> LDS (18 floats each; rows past M: zeros)
    for (int i = tid; i < BM * 18; i += kThreads) {
        const int r = i / 18, c = i - r * 18;
        const int m = m0 + r;
        off_s[r][c] = m < a.M ? a.off[(size_t)m * a.off_ld + c] : 0.f;
    }

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.w[grp], 0, (int)a.w_bytes, 0x00020000);
    unsigned w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) w_off[i] = (unsigned)(((size_t)(n0 + r0 + 32 * i) * a.k_pad + cc * 4) * 4);

    // pixel coordinates of this thread's two rows
    int ph[2], pw[2];
    unsigned pbase[2];             // byte offset of (image b, channel chunk of this thread) -- the corner's pixel offset is added
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + r0 + 32 * i;
        pok[i] = m < a.M;
        const int mm = pok[i] ? m : 0;
        const unsigned t2 = (unsigned)mm / (unsigned)a.W;
        pw[i] = (int)((unsigned)mm - t2 * a.W);
        const unsigned b = t2 / (unsigned)a.H;
        ph[i] = (int)(t2 - b * a.H);
        pbase[i] = (unsigned)(((size_t)b * a.H * a.W * a.C + (size_t)grp * a.cpg + cc * 4) * 4);
    }

    const int wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, g = lane >> 4;
    const int key = (l16 >> 1) & 7;
    const int fo0 = ((0 + g) ^ key) * 4, fo1 = ((4 + g) ^ key) * 4;
    const int w_frag = (16 * wave + l16) * BK;
    const int x_frag = l16 * BK;

    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-tap sampling state of the two rows: corner byte offsets (relative to the image's first pixel; 0xffffffff = outside)
    // and bilinear weights
    unsigned co[2][4];
    float cw[2][4];
    f32x4 rx[2][4], rw[2];
    f32x4 fw0, fw1, fx0[MB], fx1[MB];
    const int kt_per_tap = a.cpg / BK;
    const int nkt = 9 * kt_per_tap;
    int ld_kt = 0, ld_tap = 0, ld_c = 0;      // next k-tile to fetch: its tap and channel chunk inside the group

    __syncthreads();                           // off_s

#define DCN_TAP_PARAMS()                                                                                  \
    do {                                                                                                  \
        const int ky_ = ld_tap / 3, kx_ = ld_tap - ky_ * 3;                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
            const float oy_ = off_s[r0 + 32 * i][2 * ld_tap], ox_ = off_s[r0 + 32 * i][2 * ld_tap + 1];   \
            const float hf_ = (float)(ph[i] - 1 + ky_) + oy_;                                             \
            const float wf_ = (float)(pw[i] - 1 + kx_) + ox_;                                             \
            const bool in_ = pok[i] && hf_ > -1.f && wf_ > -1.f && hf_ < (float)a.H && wf_ < (float)a.W;  \
            const int hl_ = (int)floorf(hf_), wl_ = (int)floorf(wf_);                                     \
            const int hh_ = hl_ + 1, wh_ = wl_ + 1;                                                       \
            const float lh_ = hf_ - (float)hl_, lw_ = wf_ - (float)wl_;                                   \
            const float uh_ = 1.f - lh_, uw_ = 1.f - lw_;                                                 \
            const bool k1_ = in_ && hl_ >= 0 && wl_ >= 0, k2_ = in_ && hl_ >= 0 && wh_ <= a.W - 1;        \
            const bool k3_ = in_ && hh_ <= a.H - 1 && wl_ >= 0, k4_ = in_ && hh_ <= a.H - 1 && wh_ <= a.W - 1; \
            cw[i][0] = k1_ ? uh_ * uw_ : 0.f; cw[i][1] = k2_ ? uh_ * lw_ : 0.f;                           \
            cw[i][2] = k3_ ? lh_ * uw_ : 0.f; cw[i][3] = k4_ ? lh_ * lw_ : 0.f;                           \
            const unsigned cb_ = (unsigned)a.C * 4u;                                                      \
            co[i][0] = k1_ ? pbase[i] + (unsigned)(hl_ * a.W + wl_) * cb_ : 0xffffffffu;                  \
            co[i][1] = k2_ ? pbase[i] + (unsigned)(hl_ * a.W + wh_) * cb_ : 0xffffffffu;                  \
            co[i][2] = k3_ ? pbase[i] + (unsigned)(hh_ * a.W + wl_) * cb_ : 0xffffffffu;                  \
            co[i][3] = k4_ ? pbase[i] + (unsigned)(hh_ * a.W + wh_) * cb_ : 0xffffffffu;                  \
        }                                                                                                 \
    } while (0)
    // requests the four corners of both rows for channel chunk ld_c of tap ld_tap, and the weight rows of k-tile ld_kt
#define DCN_LOAD()                                                                                        \
    do {                                                                                                  \
        if (ld_c == 0) DCN_TAP_PARAMS();                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                 \
                rx[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, co[i][c], ld_c * (BK * 4), 0)); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[i], ld_kt * (BK * 4), 0)); \
        ++ld_kt;                                                                                          \
        if (++ld_c == kt_per_tap) { ld_c = 0; ++ld_tap; }                                                 \
    } while (0)
    // bilinear combination (the order of deform_im2col3x3_kernel) and the store of both operand tiles
#define DCN_STORE(BUF)                                                                                    \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
            f32x4 v_;                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                 \
                v_[e] = cw[i][0] * rx[i][0][e] + cw[i][1] * rx[i][1][e] + cw[i][2] * rx[i][2][e] + cw[i][3] * rx[i][3][e]; \
            *reinterpret_cast<f32x4 *>(Xs0 + (BUF) * kBuf + (r0 + 32 * i) * BK + cs * 4) = v_;            \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
            *reinterpret_cast<f32x4 *>(Ws0 + (BUF) * kBuf + (r0 + 32 * i) * BK + cs * 4) = rw[i];          \
    } while (0)
#define DCN_READ(FW, FX, BUF, FO)                                                                         \
    do {                                                                                                  \
        FW = *reinterpret_cast<const f32x4 *>(Ws0 + (BUF) * kBuf + w_frag + (FO));                        \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                 \
            FX[mb] = *reinterpret_cast<const f32x4 *>(Xs0 + (BUF) * kBuf + x_frag + mb * 16 * BK + (FO)); \
    } while (0)
#define DCN_MFMA(FW, FX, J0, J1)                                                                          \
    do {                                                                                                  \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j)                                               \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                             \
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(FW[j], FX[mb][j], acc[mb], 0, 0, 0);       \
    } while (0)
#define DCN_SB() __builtin_amdgcn_sched_barrier(0)
    // (one register stage: the tile requested at the top of a phase is combined and stored in the same phase, so the corner
    //  weights cw computed for that request are still the current ones at the store)
#define DCN_PHASE(BUF, HAVE_NEXT)                                                                         \
    do {                                                                                                  \
        if (HAVE_NEXT) DCN_LOAD();                                                                                       \
        DCN_READ(fw1, fx1, BUF, fo1);                                                                     \
        DCN_SB();                                                                                         \
        DCN_MFMA(fw0, fx0, 0, 4);                                                                         \
        DCN_SB();                                                                                         \
        DCN_MFMA(fw1, fx1, 0, 2);      /* (the tile requested above gets three quarters of the phase to arrive) */ \
        DCN_SB();                                                                                         \
        if (HAVE_NEXT) {                                                                                            \
            DCN_STORE((BUF) ^ 1);                                                                         \
            __syncthreads();                                                                              \
            DCN_READ(fw0, fx0, (BUF) ^ 1, fo0);                                                           \
        }                                                                                                 \
        DCN_SB();                                                                                         \
        DCN_MFMA(fw1, fx1, 2, 4);                                                                         \
        DCN_SB();                                                                                         \
    } while (0)

    DCN_LOAD();
    DCN_STORE(0);
    __syncthreads();
    DCN_READ(fw0, fx0, 0, fo0);
    for (int kt = 0; kt < nkt; kt += 2) {          // (nkt = 9 * cpg / 32 may be odd)
        DCN_PHASE(0, kt + 1 < nkt);
        if (kt + 1 >= nkt) break;
        DCN_PHASE(1, kt + 2 < nkt);
    }
#undef DCN_TAP_PARAMS
#undef DCN_LOAD
#undef DCN_STORE
#undef DCN_READ
#undef DCN_MFMA
#undef DCN_SB
#undef DCN_PHASE

    // accumulator tile mb: pixel m0 + 16 mb + l16, output channels (group grp) n0 + 16 wave + 4 g + (0..3)
    const int col = n0 + 16 * wave + 4 * g;
    if (col < a.opg) {
        float *yb = a.y + (size_t)(m0 + l16) * a.y_ld + a.y_coff + grp * a.opg + col;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            if (m0 + 16 * mb + l16 < a.M) *reinterpret_cast<f32x4 *>(yb + (size_t)mb * 16 * a.y_ld) = acc[mb];
    }
}

}  // namespace

// x f32 NHWC [B, H, W, C]; offset f32 [B, H, W, off_ld] with (dy, dx) of tap t at 2 t, 2 t + 1 (mmcv's layout); w_packed[g]:
// the group's weights [opg, 9 * cpg] (k = tap * cpg + ci) packed by sgv3d_conv_pack_weight as a 1x1 layer -> [cout_pad][k_pad];
// y f32 [B, H, W, y_ld], channels [y_coff, y_coff + groups * opg) written.  cpg % 32 == 0, opg % 4 == 0, groups <= 8.
extern "C" int sgv3d_deform_conv3x3_forward(int batch, int h, int w, int channels, int groups, int out_per_group, const float *x,
                                            const float *offset, int off_ld, const float *const *w_packed, int k_pad, int cout_pad,
                                            float *y, int y_ld, int y_coff, void *stream) {
    SGV3D_REQUIRE(batch > 0 && h > 0 && w > 0 && channels > 0 && groups > 0 && groups <= kMaxGroups && channels % groups == 0 &&
                      out_per_group > 0 && off_ld >= 18,
                  "deform_conv3x3_forward: bad shape");
    const int cpg = channels / groups;
    SGV3D_REQUIRE(cpg % BK == 0 && out_per_group % 4 == 0 && k_pad >= 9 * cpg && k_pad % BK == 0 && cout_pad >= out_per_group &&
                      cout_pad % 64 == 0 && (y_ld & 3) == 0 && (y_coff & 3) == 0 && y_ld >= y_coff + groups * out_per_group,
                  "deform_conv3x3_forward: channels per group %% 32, outputs per group %% 4, packed geometry, output stride / offset %% 4 "
                  "(cpg=%d opg=%d k_pad=%d cout_pad=%d y_ld=%d y_coff=%d)", cpg, out_per_group, k_pad, cout_pad, y_ld, y_coff);
    SGV3D_REQUIRE(x && offset && w_packed && y, "deform_conv3x3_forward: null pointer");
    const long long M = (long long)batch * h * w;
    SGV3D_REQUIRE(M < 0x7fffffffLL && M * channels * 4 < 0xf0000000LL && (long long)cout_pad * k_pad * 4 < 0xf0000000LL,
                  "deform_conv3x3_forward: tensors larger than 3.75 GiB (32-bit buffer offsets)");
    DcnArgs a;
    a.x = x; a.off = offset; a.y = y;
    uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y);
    for (int g = 0; g < kMaxGroups; ++g) {
        a.w[g] = g < groups ? w_packed[g] : nullptr;
        SGV3D_REQUIRE(g >= groups || a.w[g] != nullptr, "deform_conv3x3_forward: null weight pointer");
        align |= reinterpret_cast<uintptr_t>(a.w[g]);
    }
    SGV3D_REQUIRE((align & 15) == 0, "deform_conv3x3_forward: pointers must be 16-B aligned");
    a.H = h; a.W = w; a.C = channels; a.cpg = cpg; a.opg = out_per_group; a.groups = groups;
    a.M = (int)M; a.off_ld = off_ld; a.y_ld = y_ld; a.y_coff = y_coff; a.k_pad = k_pad;
    a.tiles_m = cdiv(M, 64);
    a.tiles_ng = groups * cdiv(out_per_group, 64);
    a.x_bytes = (unsigned)(M * channels * 4);
    a.w_bytes = (unsigned)((long long)cout_pad * k_pad * 4);
    SGV3D_REQUIRE((long long)a.tiles_m * a.tiles_ng < 0x7fffffffLL, "deform_conv3x3_forward: too many tiles");
    hipLaunchKernelGGL(dcn3x3_fused_kernel, dim3(a.tiles_m * a.tiles_ng), dim3(kThreads), 0, as_stream(stream), a);
    return check_launch("dcn3x3_fused_kernel");
}
