// Grouped GEMM with float32-accurate products on the bf16 matrix cores ("f32x3") for the middle launch of the three-launch
// Winograd F(4x4, 3x3) path (conv_wino4.hip) -- the 3x3 / stride 1 convolutions with >= 128 channels on small maps that the
// reference runs through cuDNN: HeightNet's ten 512 -> 512 layers at 54x96 (layers/backbones/lss_fpn.py:186-198), ResNet
// layer 2-4 conv2, the BEV trunk (layers/heads/bev_height_head.py:97-108).
//
//   M[p][t][co] = sum_ci V[p][t][ci] * U[p][co][ci]        p = 0..35 positions, t = Winograd tiles (rows)
//
// Arithmetic.  Every f32 operand x is split EXACTLY into three bf16 terms, hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi -
// mid) (the remainders are exact in f32: 3 x 8 significand bits = f32's 24), by the kernels that PRODUCE the operands -- the
// weight packer once per parameter version, the input transform (a bandwidth-bound kernel) per call -- so this kernel holds no
// conversion instruction.  A product v * u is the f32 sum of the six partial products of weight >= 2^-16,
//   lo_u hi_v + hi_u lo_v + mid_u mid_v + mid_u hi_v + hi_u mid_v + hi_u hi_v,
// accumulated in f32 by the MFMA; the three dropped ones are below 2^-24 of the product: one f32 rounding, as
// v_mfma_f32_16x16x4_f32 would commit.  Six v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) do the work of eight
// v_mfma_f32_16x16x4_f32 (32 cycles each): 96 cycles against 256.
//
// Layouts (bf16, "plane-interleaved k-chunks"): a row of an operand is K / 32 records of 192 bytes, record = [hi | mid | lo] x 32
// consecutive k.  One k-step of a row is one contiguous 192-byte read.
//   V3  [36][rows][K/32][3][32]      written by wino4_input_x3_kernel        rows = tiles padded to the m-tile (16 MA)
//   U3  [36][cout_pad/16][K/32][3][512]  written by wino4_pack_weight_x3_kernel, FRAGMENT order: block of 16 output channels x
//       k-step x plane = 1024 contiguous bytes, element (channel c, k) at ((k / 8) * 16 + c) * 8 + k % 8 -- the 64 lanes of a
//       wave load one MFMA operand with one buffer_load_dwordx4 each, contiguously; cout_pad = cout rounded up to 32
//   M   [36][rows][N] f32            read by wino4_output_kernel (unchanged)
//
// Kernel.  Workgroup = WN waves; tile (16 MA) rows x (32 WN) columns x 32 k; wave w owns columns [32 w, 32 w + 32) and all rows:
// MA x 2 accumulator tiles of 4 registers.  C^T = U . V^T (the weight fragment is the MFMA's A operand) so that a lane ends up
// with 4 consecutive output channels of one row: one 16-byte store per accumulator tile.  Per k-step a wave reads 6 weight
// fragments once and 3 row fragments per row block (ds_read_b128 each) for 12 MA MFMAs: (3 MA + 6) / (12 MA) reads per MFMA --
// 0.32 at MA = 7, a third of what saturates the LDS array beside 16-cycle MFMAs.
//   weights  never touch LDS: a wave's 6 fragments of a k-step (2 column blocks x 3 planes, 1 KB each) come from L2 straight into
//            registers with one buffer_load_dwordx4 per lane and fragment, requested one k-step ahead.
//   LDS      the ROW tile only, two buffers of three planes; rows of 64 bytes (32 k), the 16-byte chunk c of row r stored at
//            c ^ (-(r >> 2) & 3): the four 16-lane groups of a ds_read_b128 (rows {0-3, 12-15} at one chunk with rows {4-11} at the
//            next, and so on) each fall on 64 distinct banks; a plane is BM x 64 + 64 bytes so that the 8-lane groups of a staging
//            store that straddle two planes or two rows use both halves of the banks.  2 x 3 x (16 MA x 64 + 64) B: 43 KB at MA 7.
//   pipeline the rows of k-step k + 1 are in registers while k-step k computes, stored into the OTHER buffer behind its MFMAs: ONE
//            barrier per k-step.  Two workgroups per CU (launch bound), sched_barriers keep the request / MFMA / store order.
//            (Prefetch distance 2, three workgroups per CU, 64-thread workgroups: measured, no gain -- DESIGN 3.9.)
//   traffic  a k-step loads 16 MA x 192 B of rows per workgroup and 32 WN x 192 B of weights (registers) for 12 MA x WN x 8192 MACs:
//            10 MACs per byte at MA 7, WN 4 -- 1.5 x the bytes of the f32 kernel for 2.7 x its MFMA rate.
//   mapping  XCD-aware: the workgroups of one XCD walk consecutive tiles (m fastest, then n, then position): the tiles that share
//            a weight panel or a row panel run on the same L2.
//
// Bound: MFMA bf16 (2.5 PFLOP/s) with executed work 6 x 2 x 36 x rows x cin x cout bf16 flop per launch.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GX3Args {
    const void *x, *w;     // V3, U3
    float *y;              // M
    int rows;              // rows per position (a multiple of 16 MA)
    int K, N;              // K % 32 == 0
    int tiles_m, tiles_n;  // per position
    int cout_pad;          // rows of one weight block
    unsigned x_bytes, w_bytes;   // whole V3; the rows of ONE weight block this launch may read
    unsigned long long w_stride; // bytes between the weight blocks of consecutive positions
};

template <int MA, int WN>
__global__ __launch_bounds__(64 * WN, 2) void gemm_x3_grouped_kernel(const GX3Args a) {
    constexpr int NT = 64 * WN, BM = 16 * MA, BN = 32 * WN;
    constexpr int PLANE = BM * 64 + 64;            // one plane of the row tile; + 64: see the header
    constexpr int BUF = 3 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    // XCD-aware tile mapping (bijective for any tile count)
    const int per_pos = a.tiles_m * a.tiles_n;
    const int ntiles = 36 * per_pos;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int p = (int)((unsigned)logical / (unsigned)per_pos);
    const int rem = logical - p * per_pos;
    const int tn = (int)((unsigned)rem / (unsigned)a.tiles_m);
    const int tm = rem - tn * a.tiles_m;
    const int m0 = p * a.rows + tm * BM, n0 = tn * BN;
    const int kb = a.K >> 5;                                   // k-steps

    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(static_cast<const unsigned char *>(a.w)) + (size_t)p * a.w_stride, 0, (int)a.w_bytes, 0x00020000);

    // ---- rows (V3): global -> registers -> LDS.  A pass moves one PLANE of NT / 4 rows: lane (tid >> 2, tid & 3) the 16-byte chunk
    // tid & 3 of row (tid >> 2) + RP pass.  One per-thread base; pass / plane / k-step go into the load's scalar offset and the
    // store's immediate (the swizzle key (row >> 2) & 3 does not depend on the pass: RP % 16 == 0).
    constexpr int RP = NT / 4, PA = (BM + RP - 1) / RP;
    static_assert(RP % 16 == 0, "a pass is a multiple of 16 rows");
    const int srow = tid >> 2, sch = tid & 3;
    const unsigned row_b = (unsigned)kb * 192u;                                   // bytes per row
    const unsigned xg = (unsigned)((long long)(m0 + srow) * kb * 192 + sch * 16);
    const unsigned sl = (unsigned)(srow * 64 + ((sch ^ ((-(srow >> 2)) & 3)) << 4));
    bool a_on[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) a_on[i] = (srow & ~15) + RP * i < BM;           // wave-uniform: a wave covers 16 rows

    // ---- weights (U3, fragment order): global -> registers, no LDS -- a wave owns its 32 columns, nobody else reads them.
    // Block cb = 16 output channels; (cb, k-step, plane) = 1024 contiguous bytes, lane l at 16 l.
    const int wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, g = lane >> 4;
    unsigned wgo[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int cb = (n0 >> 4) + 2 * wave + nb;
        wgo[nb] = (cb * 16 < a.cout_pad) ? (unsigned)(((long long)cb * kb) * 3072 + lane * 16) : 0xffffffffu;
    }
    const unsigned sw = (unsigned)((g ^ ((-(l16 >> 2)) & 3)) << 4);
    const unsigned x_frag = (unsigned)(l16 * 64) + sw;                               // + 16 ma rows, + plane, + buffer

    f32x4 acc[MA][2];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4 rx[3][PA];
    bf16x8 fw[2][2][3];                             // [stage][nb][plane]
    const int nkt = kb;

#define GX3_LOAD_X(KT)                                                                                    \
    do {                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                     \
            _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                \
                rx[s][i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(                \
                    x_rsrc, a_on[i] ? xg : 0xffffffffu, (int)(row_b * (unsigned)(RP * i)) + (KT) * 192 + s * 64, 0)); \
    } while (0)
#define GX3_STORE_X(B)                                                                                    \
    do {                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                     \
            _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                \
                if (a_on[i]) *reinterpret_cast<i32x4 *>(smem + (B) * BUF + sl + s * PLANE + i * RP * 64) = rx[s][i]; \
    } while (0)
#define GX3_LOAD_W(KT, ST)                                                                                \
    do {                                                                                                  \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                  \
            _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                 \
                fw[ST][nb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wgo[nb], (KT) * 3072 + s * 1024, 0)); \
    } while (0)
#define GX3_FRAG(OFF) (*reinterpret_cast<const bf16x8 *>(smem + (OFF)))
#define GX3_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#define GX3_SB() __builtin_amdgcn_sched_barrier(0)
    // One k-step: MFMAs from LDS buffer B and weight stage ST.  The weights of the next k-step are requested at the top; the rows of
    // the next k-step (in registers since the previous phase) go to the other LDS buffer behind all but the last block's MFMAs, the
    // rows of the k-step after that are requested right behind that store; one barrier per k-step.
#define GX3_PHASE(KT, B, ST)                                                                              \
    do {                                                                                                  \
        if ((KT) + 1 < nkt) GX3_LOAD_W((KT) + 1, (ST) ^ 1);                                               \
        bf16x8 fx[2][3];                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) fx[0][s] = GX3_FRAG((B) * BUF + x_frag + s * PLANE); \
        _Pragma("unroll") for (int ma = 0; ma < MA; ++ma) {                                               \
            if (ma + 1 < MA) {                                                                            \
                _Pragma("unroll") for (int s = 0; s < 3; ++s)                                             \
                    fx[(ma + 1) & 1][s] = GX3_FRAG((B) * BUF + x_frag + (ma + 1) * 16 * 64 + s * PLANE);  \
            }                                                                                             \
            if (ma == (MA > 2 ? MA - 2 : MA - 1) && (KT) + 1 < nkt) {                                     \
                GX3_STORE_X((B) ^ 1);                                                                     \
                if ((KT) + 2 < nkt) GX3_LOAD_X((KT) + 2);                                                 \
            }                                                                                             \
            GX3_SB();                                                                                     \
            const bf16x8 *v = fx[ma & 1];                                                                 \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                            \
                /* small terms first: (u lo, v hi), (u hi, v lo), (mid, mid), (u mid, v hi), (u hi, v mid), (hi, hi) */ \
                GX3_MFMA(fw[ST][nb][2], v[0], acc[ma][nb]);                                               \
                GX3_MFMA(fw[ST][nb][0], v[2], acc[ma][nb]);                                               \
                GX3_MFMA(fw[ST][nb][1], v[1], acc[ma][nb]);                                               \
                GX3_MFMA(fw[ST][nb][1], v[0], acc[ma][nb]);                                               \
                GX3_MFMA(fw[ST][nb][0], v[1], acc[ma][nb]);                                               \
                GX3_MFMA(fw[ST][nb][0], v[0], acc[ma][nb]);                                               \
            }                                                                                             \
            GX3_SB();                                                                                     \
        }                                                                                                 \
        if ((KT) + 1 < nkt) __syncthreads();                                                              \
    } while (0)

    GX3_LOAD_X(0);
    GX3_LOAD_W(0, 0);
    GX3_STORE_X(0);
    if (nkt > 1) GX3_LOAD_X(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        GX3_PHASE(kt, 0, 0);
        GX3_PHASE(kt + 1, 1, 1);
    }
    if (kt < nkt) GX3_PHASE(kt, 0, 0);
#undef GX3_LOAD_X
#undef GX3_STORE_X
#undef GX3_LOAD_W
#undef GX3_FRAG
#undef GX3_MFMA
#undef GX3_SB
#undef GX3_PHASE

    // accumulator tile (ma, nb): row (Winograd tile) m0 + 16 ma + l16, columns n0 + 32 wave + 16 nb + 4 g + (0..3)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + 32 * wave + 16 * nb + 4 * g;
        if (col < a.N) {
            float *yb = a.y + (size_t)(m0 + l16) * a.N + col;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) *reinterpret_cast<f32x4 *>(yb + (size_t)ma * 16 * a.N) = acc[ma][nb];
        }
    }
}

template <int MA, int WN>
int launch_x3(const GX3Args &a, hipStream_t st) {
    hipLaunchKernelGGL((gemm_x3_grouped_kernel<MA, WN>), dim3(36 * a.tiles_m * a.tiles_n), dim3(64 * WN), 0, st, a);
    return check_launch("gemm_x3_grouped_kernel");
}

}  // namespace

namespace sgv3d {

// m-tile heights (in 16-row blocks) and wave counts of the instantiations; variant = index into kX3Ma + 5 * (WN == 5)
static const int kX3Ma[5] = {3, 4, 6, 7, 8};

int gemm_x3_tile_rows(int variant) { return variant >= 0 && variant < 10 ? 16 * kX3Ma[variant % 5] : 0; }

// x: V3 [36][rows][K/32][3][32] bf16, w: U3 36 blocks of [cout_pad][K/32][3][32] bf16 w_block_stride bytes apart (a chunk of
// output channels reads the tail of every block), y: M [36][rows][N] f32.  rows % (16 MA) == 0, K % 32 == 0, N % 4 == 0.
int conv_gemm_grouped_x3(const void *x, const void *w, float *y, int rows, int K, int N, int cout_pad, int variant, hipStream_t st,
                         size_t w_block_stride) {
    SGV3D_REQUIRE(x && w && y && rows > 0 && K > 0 && N > 0 && variant >= 0 && variant < 10, "conv_gemm_grouped_x3: bad argument");
    const int ma = kX3Ma[variant % 5], wn = variant >= 5 ? 5 : 4;
    SGV3D_REQUIRE(rows % (16 * ma) == 0 && K % 32 == 0 && N % 4 == 0 && cout_pad >= N && w_block_stride >= (size_t)cout_pad * K * 6,
                  "conv_gemm_grouped_x3: rows %% %d, K %% 32, N %% 4, cout_pad >= N (rows=%d K=%d N=%d cout_pad=%d)", 16 * ma, rows, K, N, cout_pad);
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  "conv_gemm_grouped_x3: pointers must be 16-B aligned");
    const long long xb = 36LL * rows * K * 6, wb = (long long)cout_pad * K * 6;
    SGV3D_REQUIRE(xb < 0xf0000000LL && wb < 0xf0000000LL && 36LL * rows < 0x7fffffffLL,
                  "conv_gemm_grouped_x3: operands larger than 3.75 GiB (32-bit buffer offsets)");
    GX3Args a;
    a.x = x; a.w = w; a.y = y;
    a.rows = rows; a.K = K; a.N = N; a.cout_pad = cout_pad;
    a.tiles_m = rows / (16 * ma);
    a.tiles_n = cdiv(N, 32 * wn);
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.w_stride = w_block_stride;
    SGV3D_REQUIRE(36LL * a.tiles_m * a.tiles_n < 0x7fffffffLL, "conv_gemm_grouped_x3: too many tiles");
    switch (variant) {
        case 0: return launch_x3<3, 4>(a, st);
        case 1: return launch_x3<4, 4>(a, st);
        case 2: return launch_x3<6, 4>(a, st);
        case 3: return launch_x3<7, 4>(a, st);
        case 4: return launch_x3<8, 4>(a, st);
        case 5: return launch_x3<3, 5>(a, st);
        case 6: return launch_x3<4, 5>(a, st);
        case 7: return launch_x3<6, 5>(a, st);
        case 8: return launch_x3<7, 5>(a, st);
        default: return launch_x3<8, 5>(a, st);
    }
}

}  // namespace sgv3d
