// Grouped f32 GEMM on v_mfma_f32_16x16x4_f32 for the middle launch of the three-launch Winograd F(4x4, 3x3) path
// (conv_wino4.hip) -- the 3x3 / stride 1 convolutions with >= 128 channels on small maps that the reference runs through
// cuDNN: HeightNet's ten 512 -> 512 layers at 54x96 (layers/backbones/lss_fpn.py:186-198), ResNet layer 3 / 4 conv2, the BEV
// trunk (layers/heads/bev_height_head.py:97-108).
//
//   M[p][t][co] = sum_ci V[p][t][ci] * U[p][co][ci]        p = 0..35 positions, t = Winograd tiles (rows), all k-contiguous
//
// Why a second GEMM kernel beside conv_igemm_kernel.  That kernel's MFMA tile is 32 x 32 (v_mfma_f32_32x32x2_f32), so the rows
// of a position are padded to a multiple of 64 (32 for its narrow form): a 54x96 map has 14 x 24 = 336 tiles -> 384 rows, 14 %
// of the GEMM multiplies padding.  v_mfma_f32_16x16x4_f32 has the same f32 rate (1024 MACs in 32 cycles against 2048 in 64)
// and a 16-row granularity: a workgroup tile of 48 rows x 64 columns covers 336 = 7 x 48 rows exactly, and 36 x 7 x 8 = 2016
// workgroups spread over 256 CUs with a smaller last round than 1728 tiles of 64 x 64.
//
//   workgroup   256 threads = 4 waves; tile (16 MB) rows x 64 columns x 32 k, MB = 3 (48 rows).  Wave w owns columns
//               [16 w, 16 w + 16) and ALL rows: MB accumulator tiles of 4 registers.
//   operands    C^T = U . V^T: the A operand is the weight fragment (16 output channels on M), B the tile fragment (16 rows on
//               N), so a lane ends up with 4 CONSECUTIVE output channels of one row -> one 16-byte store per accumulator tile.
//   k order     lane group g = lane / 16 supplies k = 16 q + 4 g + j for MFMA j of k-group q in BOTH operands: one
//               ds_read_b128 per fragment feeds four MFMAs (a permutation of the k sum only; f32 rounding order).
//   LDS         rows of 32 floats (128 B), no padding, the 16-byte chunk stored at chunk ^ ((row >> 1) & 7): the 16 lanes of a
//               fragment read (16 consecutive rows, one chunk) and of a staging store (2 rows x 8 chunks) fall on 16 distinct
//               16-byte slots.  2 x (48 + 64) x 128 B = 28 KB per workgroup, ~80 registers -> five workgroups per CU.
//   pipeline    one register stage, two LDS buffers, one barrier per k-tile (the scheme of the five-per-CU 64x64 tile): tile
//               t+1 is requested at the top of tile t's phase and written to the other buffer behind three quarters of its MFMAs.
//   epilogue    raw accumulators (the output transform kernel applies BN / residual / ReLU).
//
// Bound: MFMA f32 (157.3 TFLOP/s); executed work 2 x 36 x rows x cin x cout per launch.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;
constexpr int kThreads = 256;

struct G16Args {
    const float *x, *w;
    float *y;
    int rows;              // rows per group (a multiple of the tile height)
    int K, N, k_pad;
    long long wb_x;        // floats between the weight blocks of consecutive groups
    int tiles_m, tiles_n;  // tiles_m counts all groups
    unsigned x_bytes, w_bytes;
};

template <int MB>
__global__ __launch_bounds__(kThreads, 5) void gemm16_grouped_kernel(const G16Args a) {
    constexpr int BM = 16 * MB, BN = 64;
    constexpr int XCH = (BM + 31) / 32;          // 16-byte chunks of X per thread and k-tile (the last one covers rows < BM only)
    constexpr int kBuf = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) float smem[2 * kBuf];
    float *const Xs0 = smem;
    float *const Ws0 = smem + BM * BK;

    // XCD-aware tile mapping (bijective for any tile count): the workgroups of one XCD walk consecutive m-tiles of one n-tile
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tn = (int)((unsigned)logical / (unsigned)a.tiles_m);
    const int tm = logical - tn * a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const float *const w_grp = a.w + (size_t)((unsigned)m0 / (unsigned)a.rows) * (size_t)a.wb_x;

    const int tid = threadIdx.x;
    const int cc = tid & 7, r0 = tid >> 3;
    const int cs = cc ^ ((r0 >> 1) & 7);         // (rows r0 and r0 + 32 share the key)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w_grp, 0, (int)a.w_bytes, 0x00020000);
    unsigned x_off[XCH], w_off[2];
#pragma unroll
    for (int i = 0; i < XCH; ++i)
        x_off[i] = (r0 + 32 * i < BM) ? (unsigned)(((long long)(m0 + r0 + 32 * i) * a.K + cc * 4) * 4) : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 2; ++i) w_off[i] = (unsigned)(((size_t)(n0 + r0 + 32 * i) * a.k_pad + cc * 4) * 4);

    const int wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, g = lane >> 4;
    const int key = (l16 >> 1) & 7;              // (row >> 1) & 7 of every row this lane reads: the block offsets are multiples of 16
    const int fo0 = ((0 + g) ^ key) * 4, fo1 = ((4 + g) ^ key) * 4;
    const int w_frag = (16 * wave + l16) * BK;
    const int x_frag = l16 * BK;

    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 rx[XCH], rw[2];
    f32x4 fw0, fw1, fx0[MB], fx1[MB];
    int ld_kt = 0;
    const int nkt = a.K / BK;

#define G16_LOAD()                                                                                        \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i)                                                   \
            rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, x_off[i], ld_kt * (BK * 4), 0)); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[i], ld_kt * (BK * 4), 0)); \
        ++ld_kt;                                                                                          \
    } while (0)
#define G16_STORE(BUF)                                                                                    \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i)                                                   \
            if (r0 + 32 * i < BM) *reinterpret_cast<f32x4 *>(Xs0 + (BUF) * kBuf + (r0 + 32 * i) * BK + cs * 4) = rx[i]; \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
            *reinterpret_cast<f32x4 *>(Ws0 + (BUF) * kBuf + (r0 + 32 * i) * BK + cs * 4) = rw[i];          \
    } while (0)
#define G16_READ(FW, FX, BUF, FO)                                                                         \
    do {                                                                                                  \
        FW = *reinterpret_cast<const f32x4 *>(Ws0 + (BUF) * kBuf + w_frag + (FO));                        \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                 \
            FX[mb] = *reinterpret_cast<const f32x4 *>(Xs0 + (BUF) * kBuf + x_frag + mb * 16 * BK + (FO)); \
    } while (0)
#define G16_MFMA(FW, FX, J0, J1)                                                                          \
    do {                                                                                                  \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j)                                               \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                             \
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(FW[j], FX[mb][j], acc[mb], 0, 0, 0);       \
    } while (0)
#define G16_SB() __builtin_amdgcn_sched_barrier(0)
#define G16_PHASE(BUF, HAVE_NEXT)                                                                         \
    do {                                                                                                  \
        if constexpr (HAVE_NEXT) G16_LOAD();                                                                                       \
        G16_READ(fw1, fx1, BUF, fo1);                                                                     \
        G16_SB();                                                                                         \
        G16_MFMA(fw0, fx0, 0, 4);                                                                         \
        G16_SB();                                                                                         \
        G16_MFMA(fw1, fx1, 0, 2);      /* (the tile requested above gets three quarters of the phase to arrive) */ \
        G16_SB();                                                                                         \
        if constexpr (HAVE_NEXT) {                                                                                            \
            G16_STORE((BUF) ^ 1);                                                                         \
            __syncthreads();                                                                              \
            G16_READ(fw0, fx0, (BUF) ^ 1, fo0);                                                           \
        }                                                                                                 \
        G16_SB();                                                                                         \
        G16_MFMA(fw1, fx1, 2, 4);                                                                         \
        G16_SB();                                                                                         \
    } while (0)

    G16_LOAD();
    G16_STORE(0);
    __syncthreads();
    G16_READ(fw0, fx0, 0, fo0);
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
        G16_PHASE(0, true);
        G16_PHASE(1, true);
    }
    if (kt + 1 < nkt) {
        G16_PHASE(0, true);
        G16_PHASE(1, false);
    } else {
        G16_PHASE(0, false);
    }
#undef G16_LOAD
#undef G16_STORE
#undef G16_READ
#undef G16_MFMA
#undef G16_SB
#undef G16_PHASE

    // accumulator tile mb: row (tile) = 16 mb + l16, columns n0 + 16 wave + 4 g + (0..3)
    const int col = n0 + 16 * wave + 4 * g;
    if (col < a.N) {
        float *yb = a.y + (size_t)(m0 + l16) * a.N + col;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<f32x4 *>(yb + (size_t)mb * 16 * a.N) = acc[mb];
    }
}

}  // namespace

namespace sgv3d {

// x: [groups][rows][K], w: groups blocks of [cout_pad][k_pad] (as conv_gemm_grouped), y: [groups][rows][N].  rows % 48 == 0,
// K % 32 == 0 (channel-chunk-major weights), N % 4 == 0.
int conv_gemm_grouped16(const float *x, const float *w, float *y, int rows, int groups, int K, int N, int k_pad, int cout_pad,
                        hipStream_t st) {
    SGV3D_REQUIRE(x && w && y && rows > 0 && groups > 0 && K > 0 && N > 0, "conv_gemm_grouped16: bad argument");
    SGV3D_REQUIRE(rows % 48 == 0 && K % BK == 0 && N % 4 == 0 && k_pad >= K && k_pad % BK == 0 && cout_pad >= N && cout_pad % 64 == 0,
                  "conv_gemm_grouped16: rows %% 48, K %% 32, N %% 4, k_pad / cout_pad as packed (rows=%d K=%d k_pad=%d N=%d cout_pad=%d)",
                  rows, K, k_pad, N, cout_pad);
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  "conv_gemm_grouped16: pointers must be 16-B aligned");
    const long long M = (long long)rows * groups;
    SGV3D_REQUIRE(M < 0x7fffffffLL && M * K * 4 < 0xf0000000LL && (long long)cout_pad * k_pad * 4 < 0xf0000000LL,
                  "conv_gemm_grouped16: operands larger than 3.75 GiB (32-bit buffer offsets)");
    G16Args a;
    a.x = x; a.w = w; a.y = y;
    a.rows = rows; a.K = K; a.N = N; a.k_pad = k_pad;
    a.wb_x = (long long)cout_pad * k_pad;
    a.tiles_m = (int)(M / 48);
    a.tiles_n = cdiv(N, 64);
    a.x_bytes = (unsigned)(M * K * 4);
    a.w_bytes = (unsigned)((long long)cout_pad * k_pad * 4);
    SGV3D_REQUIRE((long long)a.tiles_m * a.tiles_n < 0x7fffffffLL, "conv_gemm_grouped16: too many tiles");
    hipLaunchKernelGGL((gemm16_grouped_kernel<3>), dim3(a.tiles_m * a.tiles_n), dim3(kThreads), 0, st, a);
    return check_launch("gemm16_grouped_kernel");
}

}  // namespace sgv3d
