// Pointwise (1x1, stride 1) convolution with float32-accurate products on the bf16 matrix cores ("f32x3") -- the ResNet
// bottlenecks' 1x1 layers, HeightNet's and the necks' 1x1 layers, which the reference runs through cuDNN (mmdet ResNet built at
// layers/backbones/lss_fpn.py:296-297; lss_fpn.py:175-205; layers/heads/bev_height_head.py:75-78).  These layers hold most of the
// frame's multiply-adds: on the f32 MFMA (conv_igemm_kernel, five workgroups per CU) they run at ~0.55 of its peak, and six
// v_mfma_f32_16x16x32_bf16 do the work of eight v_mfma_f32_16x16x4_f32 in 96 cycles instead of 256.
//
//   y[m][co] = act( scale[co] * sum_ci x[m][ci] * w[co][ci] + shift[co] (+ residual[m][co]) )        m = pixel (NHWC row)
//
// Arithmetic: as gemm_x3_grouped.hip -- x = hi + mid + lo exactly (three bf16 terms), the six partial products of weight >=
// 2^-16 accumulated in f32, the three dropped ones below one f32 rounding of the product.  The WEIGHTS are split once, by the
// packer (sgv3d_conv_pack_weight_x3: [cout_pad / 16][cin / 32][3][512] bf16, the fragment-ordered layout of the F(4x4) position GEMM's weights).  The
// ACTIVATIONS stay f32 in HBM (the tensors between the layers do not change) and are split on their way into LDS: a thread
// loads 4 consecutive channels of a pixel (16 bytes), forms the three bf16 quads (~24 vector instructions, issued in the
// shadow of the 16-cycle MFMAs of the other waves) and stores three 8-byte pieces.
//
// Kernel: the core of gemm_x3_grouped_kernel.  Workgroup = WN waves, tile (16 MA) pixels x (32 WN) channels x 32 k; wave w owns
// channels [32 w, 32 w + 32) and all pixels; C^T = W . X^T so that a lane ends up with 4 consecutive output channels of one
// pixel (the epilogue is one 16-byte load of scale / shift / residual and one 16-byte store per accumulator tile).  The weights go
// from global memory straight into the wave's registers (fragment order, nobody else reads a wave's channels); the pixels go
// through LDS: one image per plane, rows of 64 bytes, 16-byte chunk c of row r at c ^ (-(r >> 2) & 3) (conflict-free fragment
// reads, see gemm_x3_grouped.hip); one register stage, two LDS buffers, one barrier per k-step.
//
// Bound: MFMA bf16 (2.5 PFLOP/s) with 6 x 2 x M x cin x cout executed bf16 flop; HBM for the layers with few input channels.
#include "conv_common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct PwX3Args {
    const float *x, *scale, *bias, *res;
    const void *w;         // U3 in fragment order, k = ((ci / 32) * taps + tap) * 32 + ci % 32
    float *y;
    int M, K, N;           // output pixels, kh * kw * cin (cin % 32 == 0), output channels (N % 4 == 0)
    int x_ld, x_coff, y_ld, y_coff, res_ld, relu;
    int cout_pad, tiles_m, tiles_n, mfirst;
    unsigned x_bytes, w_bytes;
    // TAPS form (a strided / multi-tap convolution as an implicit GEMM): geometry of the gather
    int in_h, in_w, out_h, out_w, kh, kw, stride, pad, dil;
    int cin4;              // MODE 2: cin / 4
    int split_k;           // > 1: blockIdx.y owns a slice of the k-steps and stores raw partial sums to ws [split_k][M][N]
    float *ws;
};

// TAPS = false: pointwise (1x1 / stride 1): the input pixel is the output pixel.  TAPS = true: kh x kw taps (<= 32), any stride /
// padding / dilation: k-step kt covers 32 channels of ONE tap (kt = chunk * taps + tap, the packed weights' k order); which taps of
// a row's pixel fall inside the image is a bit mask made once, a k-step costs one bit test and one add per 16-byte chunk.
// MODE 2 (tap-major, the weights' k order 0: k = tap * cin + ci, cin % 4 == 0, <= 64 taps): the stems -- 7x7 / stride 2 on the 4-channel
// image and on the 80-channel BEV map.  A 16-byte chunk is 4 channels of ONE tap; the thread that owns chunk `ach` of a row walks
// its own (tap, channel) pair from k-step to k-step (q = 8 kt + ach, tap = q / (cin / 4)); validity is a 64-bit mask per row.
template <int MA, int WN, int MODE>
__global__ __launch_bounds__(64 * WN, 2) void conv_pw_x3_kernel(const PwX3Args a) {
    constexpr bool TAPS = MODE == 1, TM = MODE == 2;
    constexpr int NT = 64 * WN, BM = 16 * MA, BN = 32 * WN;
    constexpr int PLANE = BM * 64 + 64;
    constexpr int BUF = 3 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    // XCD-aware tile walk (bijective for any tile count); default: the workgroups of one XCD walk the m-tiles of one channel
    // tile (weight panel shared); mfirst: the channel tiles of one m-tile (pixel rows fetched once)
    const int ntiles = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    int tn, tm;
    if (a.mfirst) {
        tm = (int)((unsigned)logical / (unsigned)a.tiles_n);
        tn = logical - tm * a.tiles_n;
    } else {
        tn = (int)((unsigned)logical / (unsigned)a.tiles_m);
        tm = logical - tn * a.tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kb = a.K >> 5;

    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.w), 0, (int)a.w_bytes, 0x00020000);

    // ---- pixels (f32, split here): global -> registers -> three bf16 planes in LDS.  A pass covers NT / 8 rows; lane
    // (tid >> 3, tid & 7) = 4 channels of one row
    constexpr int RPA = NT / 8, PA = (BM + RPA - 1) / RPA;
    static_assert(RPA % 16 == 0, "a pass is a multiple of 16 rows");
    const int arow = tid >> 3, ach = tid & 7;
    unsigned xg[PA];
    [[maybe_unused]] unsigned xmask[PA];
    [[maybe_unused]] unsigned long long xmask64[PA];
    bool a_on[PA];
    const int taps = (TAPS || TM) ? a.kh * a.kw : 1;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = arow + RPA * i;
        a_on[i] = (arow & ~7) + RPA * i < BM;                                   // wave-uniform: a wave covers 8 rows
        const bool live = a_on[i] && m0 + r < a.M;
        if constexpr (!TAPS && !TM) {
            xg[i] = live ? (unsigned)((((long long)(m0 + r)) * a.x_ld + a.x_coff + ach * 4) * 4) : 0xffffffffu;
        } else if constexpr (TM) {
            const int m = live ? m0 + r : 0;
            const int t = (int)((unsigned)m / (unsigned)a.out_w);
            const int ow = m - t * a.out_w;
            const int n = (int)((unsigned)t / (unsigned)a.out_h);
            const int oh = t - n * a.out_h;
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            xg[i] = (unsigned)((((long long)(n * a.in_h + ih0) * a.in_w + iw0) * a.x_ld + a.x_coff) * 4);
            unsigned long long mk = 0;
            int tt = 0;
            for (int th = 0; th < a.kh; ++th) {
                const bool row_in = live && (unsigned)(ih0 + th * a.dil) < (unsigned)a.in_h;
                for (int tw = 0; tw < a.kw; ++tw, ++tt) mk |= (row_in && (unsigned)(iw0 + tw * a.dil) < (unsigned)a.in_w) ? (1ull << tt) : 0ull;
            }
            xmask64[i] = mk;
        } else {
            const int m = live ? m0 + r : 0;
            const int t = (int)((unsigned)m / (unsigned)a.out_w);
            const int ow = m - t * a.out_w;
            const int n = (int)((unsigned)t / (unsigned)a.out_h);
            const int oh = t - n * a.out_h;
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            // (the top-left tap may lie outside the image: the offset is formed in 64 bits and used only under its tap's mask bit)
            xg[i] = (unsigned)((((long long)(n * a.in_h + ih0) * a.in_w + iw0) * a.x_ld + a.x_coff + ach * 4) * 4);
            unsigned mk = 0;
            int tt = 0;
            for (int th = 0; th < a.kh; ++th) {
                const bool row_in = live && (unsigned)(ih0 + th * a.dil) < (unsigned)a.in_h;
                for (int tw = 0; tw < a.kw; ++tw, ++tt) mk |= (row_in && (unsigned)(iw0 + tw * a.dil) < (unsigned)a.in_w) ? (1u << tt) : 0u;
            }
            xmask[i] = mk;
        }
    }
    // TAPS: the walk over (channel chunk, tap) of the loads, wave-uniform
    [[maybe_unused]] int ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_c0 = 0;
    // MODE 2: this thread's own walk -- chunk q = 8 kt + ach of the k axis is channels [4 tm_c4, 4 tm_c4 + 4) of tap tm_tap = (tm_kh, tm_kw)
    [[maybe_unused]] const int cq = a.cin4;
    [[maybe_unused]] int tm_tap = 0, tm_kh = 0, tm_kw = 0, tm_c4 = 0;
    // LDS: row r, 8-byte piece ach of its 64 bytes: 16-byte chunk (ach >> 1) swizzled, half (ach & 1)
    const unsigned xl = (unsigned)(arow * 64 + ((((ach >> 1) ^ ((-(arow >> 2)) & 3)) << 4) | ((ach & 1) << 3)));

    // ---- weights (U3, fragment order): global -> registers, no LDS (a wave owns its 32 channels)
    const int wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, g = lane >> 4;
    unsigned wgo[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int cb = (n0 >> 4) + 2 * wave + nb;
        wgo[nb] = (cb * 16 < a.cout_pad) ? (unsigned)(((long long)cb * kb) * 3072 + lane * 16) : 0xffffffffu;
    }
    const unsigned sw = (unsigned)((g ^ ((-(l16 >> 2)) & 3)) << 4);
    const unsigned x_frag = (unsigned)(l16 * 64) + sw;

    f32x4 acc[MA][2];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 rx[PA];
    bf16x8 fw[2][2][3];
    // split-K: this workgroup's k-steps are [kt0, kt0 + nkt)
    int kt0 = 0, nkt = kb;
    if (a.split_k > 1) {
        kt0 = (int)((unsigned)kb * blockIdx.y / (unsigned)a.split_k);
        nkt = (int)((unsigned)kb * (blockIdx.y + 1) / (unsigned)a.split_k) - kt0;
        if constexpr (TAPS) {
            const int chunk = kt0 / taps;
            ld_tap = kt0 - chunk * taps;
            ld_c0 = chunk * 32;
            ld_kh = ld_tap / a.kw;
            ld_kw = ld_tap - ld_kh * a.kw;
        }
    }
    if constexpr (TM) {
        const int q = kt0 * 8 + ach;
        tm_tap = q / cq;
        tm_c4 = q - tm_tap * cq;
        tm_kh = tm_tap / a.kw;
        tm_kw = tm_tap - tm_kh * a.kw;
    }

    // (TAPS: called with consecutive KT = 0, 1, 2, ...: the tap walk advances by one per call)
#define PWX3_LOAD_X(KT)                                                                                   \
    do {                                                                                                  \
        if constexpr (!TAPS && !TM) {                                                                     \
            _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                \
                rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xg[i], (kt0 + (KT)) * 128, 0)); \
        } else if constexpr (TM) {                                                                        \
            const unsigned koff_ = (unsigned)((((tm_kh * a.in_w + tm_kw) * a.dil) * a.x_ld + tm_c4 * 4) * 4); \
            const unsigned long long bit_ = tm_tap < taps ? 1ull << tm_tap : 0ull;     /* (the zero padding of K) */ \
            _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                \
                rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                   \
                    x_rsrc, (xmask64[i] & bit_) ? xg[i] + koff_ : 0xffffffffu, 0, 0));                    \
            tm_c4 += 8;                                                                                   \
            while (tm_c4 >= cq) {                                                                         \
                tm_c4 -= cq;                                                                              \
                ++tm_tap;                                                                                 \
                if (++tm_kw == a.kw) { tm_kw = 0; ++tm_kh; }                                              \
            }                                                                                             \
        } else {                                                                                          \
            const unsigned koff_ = (unsigned)((((ld_kh * a.in_w + ld_kw) * a.dil) * a.x_ld + ld_c0) * 4);  \
            const unsigned bit_ = 1u << ld_tap;                                                           \
            _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                \
                rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                   \
                    x_rsrc, (xmask[i] & bit_) ? xg[i] + koff_ : 0xffffffffu, 0, 0));                      \
            ++ld_tap;                                                                                     \
            if (++ld_kw == a.kw) {                                                                        \
                ld_kw = 0;                                                                                \
                if (++ld_kh == a.kh) { ld_kh = 0; ld_tap = 0; ld_c0 += 32; }                              \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
#define PWX3_STORE_X(B)                                                                                   \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                    \
            if (a_on[i]) {                                                                                \
                const bf16x4 hi = __builtin_convertvector(rx[i], bf16x4);                                 \
                const f32x4 r1 = rx[i] - __builtin_convertvector(hi, f32x4);                              \
                const bf16x4 mid = __builtin_convertvector(r1, bf16x4);                                   \
                const f32x4 r2 = r1 - __builtin_convertvector(mid, f32x4);                                \
                const bf16x4 lo = __builtin_convertvector(r2, bf16x4);                                    \
                unsigned char *d = smem + (B) * BUF + xl + i * RPA * 64;                                  \
                *reinterpret_cast<bf16x4 *>(d) = hi;                                                      \
                *reinterpret_cast<bf16x4 *>(d + PLANE) = mid;                                             \
                *reinterpret_cast<bf16x4 *>(d + 2 * PLANE) = lo;                                          \
            }                                                                                             \
    } while (0)
#define PWX3_LOAD_W(KT, ST)                                                                               \
    do {                                                                                                  \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                  \
            _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                 \
                fw[ST][nb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wgo[nb], (kt0 + (KT)) * 3072 + s * 1024, 0)); \
    } while (0)
#define PWX3_FRAG(OFF) (*reinterpret_cast<const bf16x8 *>(smem + (OFF)))
#define PWX3_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#define PWX3_SB() __builtin_amdgcn_sched_barrier(0)
    // one k-step (as gemm_x3_grouped.hip): weights of the next k-step requested at the top; the pixels of the next k-step (in
    // registers since the previous phase) are split and stored to the other LDS buffer behind all but the last block's MFMAs, the
    // pixels of the k-step after that are requested right behind that store; one barrier per k-step
#define PWX3_PHASE(KT, B, ST)                                                                             \
    do {                                                                                                  \
        if ((KT) + 1 < nkt) PWX3_LOAD_W((KT) + 1, (ST) ^ 1);                                              \
        bf16x8 fx[2][3];                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) fx[0][s] = PWX3_FRAG((B) * BUF + x_frag + s * PLANE); \
        _Pragma("unroll") for (int ma = 0; ma < MA; ++ma) {                                               \
            if (ma + 1 < MA) {                                                                            \
                _Pragma("unroll") for (int s = 0; s < 3; ++s)                                             \
                    fx[(ma + 1) & 1][s] = PWX3_FRAG((B) * BUF + x_frag + (ma + 1) * 16 * 64 + s * PLANE); \
            }                                                                                             \
            if (ma == (MA > 2 ? MA - 2 : MA - 1) && (KT) + 1 < nkt) {                                     \
                PWX3_STORE_X((B) ^ 1);                                                                    \
                if ((KT) + 2 < nkt) PWX3_LOAD_X((KT) + 2);                                                \
            }                                                                                             \
            PWX3_SB();                                                                                    \
            const bf16x8 *v = fx[ma & 1];                                                                 \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                            \
                PWX3_MFMA(fw[ST][nb][2], v[0], acc[ma][nb]);                                              \
                PWX3_MFMA(fw[ST][nb][0], v[2], acc[ma][nb]);                                              \
                PWX3_MFMA(fw[ST][nb][1], v[1], acc[ma][nb]);                                              \
                PWX3_MFMA(fw[ST][nb][1], v[0], acc[ma][nb]);                                              \
                PWX3_MFMA(fw[ST][nb][0], v[1], acc[ma][nb]);                                              \
                PWX3_MFMA(fw[ST][nb][0], v[0], acc[ma][nb]);                                              \
            }                                                                                             \
            PWX3_SB();                                                                                    \
        }                                                                                                 \
        if ((KT) + 1 < nkt) __syncthreads();                                                              \
    } while (0)

    PWX3_LOAD_X(0);
    PWX3_LOAD_W(0, 0);
    PWX3_STORE_X(0);
    if (nkt > 1) PWX3_LOAD_X(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        PWX3_PHASE(kt, 0, 0);
        PWX3_PHASE(kt + 1, 1, 1);
    }
    if (kt < nkt) PWX3_PHASE(kt, 0, 0);
#undef PWX3_LOAD_X
#undef PWX3_STORE_X
#undef PWX3_LOAD_W
#undef PWX3_FRAG
#undef PWX3_MFMA
#undef PWX3_SB
#undef PWX3_PHASE

    // epilogue: accumulator tile (ma, nb) of this lane = pixel m0 + 16 ma + l16, channels n0 + 32 wave + 16 nb + 4 g + (0..3)
    if (a.split_k > 1) {                 // raw partial sums; conv_splitk_reduce4_kernel adds them in fixed order and runs the epilogue
        float *wsb = a.ws + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = n0 + 32 * wave + 16 * nb + 4 * g;
            if (col >= a.N) continue;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
                const int row = m0 + 16 * ma + l16;
                if (row < a.M) *reinterpret_cast<f32x4 *>(wsb + (size_t)row * a.N + col) = acc[ma][nb];
            }
        }
        return;
    }
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + 32 * wave + 16 * nb + 4 * g;
        if (col >= a.N) continue;
        const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) {
            const int row = m0 + 16 * ma + l16;
            if (row >= a.M) continue;
            f32x4 v = acc[ma][nb] * sc + sh;
            if (a.res) v += *reinterpret_cast<const f32x4 *>(a.res + (size_t)row * a.res_ld + col);
            v = f32x4{fmaxf(v[0], floor_), fmaxf(v[1], floor_), fmaxf(v[2], floor_), fmaxf(v[3], floor_)};
            *reinterpret_cast<f32x4 *>(a.y + (size_t)row * a.y_ld + a.y_coff + col) = v;
        }
    }
}

// w [cout][cin][kh][kw] f32 (OIHW) -> U3 [cout_pad / 16][K / 32][3][512] bf16 with k = ((ci / 32) * taps + tap) * 32 + ci % 32
// (K = taps * cin_pad), zero rows / columns in the padding
// (k_order 0: k = tap * cin_pad + ci, K padded to a multiple of 32 with zeros: the stems)
__global__ __launch_bounds__(256) void pack_weight_x3_kernel(const float *__restrict__ w, int cout, int cin, int cin_pad, int cout_pad,
                                                             int taps, int k_order, __bf16 *__restrict__ u) {
    const long long kp = k_order ? (long long)taps * cin_pad : ((long long)taps * cin_pad + 31) / 32 * 32;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)cout_pad * kp) return;
    const int co = (int)(i / kp), k = (int)(i - (long long)co * kp);
    int tap, ci;
    if (k_order) {
        const int chunk = k / (32 * taps), rem = k - chunk * 32 * taps;
        tap = rem >> 5;
        ci = chunk * 32 + (rem & 31);
    } else {
        tap = k / cin_pad;
        ci = k - tap * cin_pad;
    }
    const float v = (co < cout && ci < cin && tap < taps) ? w[((size_t)co * cin + ci) * taps + tap] : 0.f;
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    // fragment order: block of 16 output channels x k-step x plane = 512 elements, (channel c, k) at ((k / 8) * 16 + c) * 8 + k % 8
    __bf16 *q = u + (((size_t)(co >> 4) * (kp >> 5) + (k >> 5)) * 3) * 512 + ((((k & 31) >> 3) * 16 + (co & 15)) * 8 + (k & 7));
    q[0] = hi;
    q[512] = mid;
    q[1024] = lo;
}

template <int MA, int WN>
int launch_pw(const PwX3Args &a, int mode, hipStream_t st) {
    const dim3 grid(a.tiles_m * a.tiles_n, a.split_k > 1 ? a.split_k : 1);
    if (mode == 2) hipLaunchKernelGGL((conv_pw_x3_kernel<MA, WN, 2>), grid, dim3(64 * WN), 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL((conv_pw_x3_kernel<MA, WN, 1>), grid, dim3(64 * WN), 0, st, a);
    else hipLaunchKernelGGL((conv_pw_x3_kernel<MA, WN, 0>), grid, dim3(64 * WN), 0, st, a);
    return check_launch("conv_pw_x3_kernel");
}

}  // namespace

// variant: m-tile {0: 32, 1: 64, 2: 128} pixels, + 4: 64 / + 8: 256 instead of 128 channels per workgroup (2 / 8 waves)
extern "C" int sgv3d_conv_pack_weight_x3(const float *w_src, int cout, int cin, int kh, int kw, int cin_pad, int cout_pad, void *u3_packed,
                                         void *stream) {
    // cin_pad % 32 == 0: channel-chunk-major k (k order 1, <= 32 taps); otherwise tap-major k (k order 0: cin_pad % 4 == 0, <= 64 taps,
    // K = taps * cin_pad rounded up to 32)
    const int k_order = cin_pad % 32 == 0 ? 1 : 0;
    SGV3D_REQUIRE(w_src && u3_packed && cout > 0 && cin > 0 && kh > 0 && kw > 0 && kh * kw <= (k_order ? 32 : 64) && cin_pad >= cin &&
                      cin_pad % 4 == 0 && cout_pad >= cout && cout_pad % 32 == 0,
                  "conv_pack_weight_x3: bad arguments (cout=%d cin=%d k=%dx%d cin_pad=%d cout_pad=%d)", cout, cin, kh, kw, cin_pad, cout_pad);
    const long long kp = k_order ? (long long)cin_pad * kh * kw : ((long long)cin_pad * kh * kw + 31) / 32 * 32;
    const long long total = (long long)cout_pad * kp;
    hipLaunchKernelGGL(pack_weight_x3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, cout, cin, cin_pad, cout_pad,
                       kh * kw, k_order, static_cast<__bf16 *>(u3_packed));
    return check_launch("pack_weight_x3_kernel");
}

extern "C" int sgv3d_conv2d_x3_forward(const sgv3d_conv_desc *d, const float *x, const void *u3_packed, const float *scale,
                                       const float *bias, const float *residual, float *y, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    SGV3D_REQUIRE(d && x && u3_packed && y, "conv2d_x3_forward: null pointer");
    const bool tapmajor = d->cin % 32 != 0;              // the weights' k order 0
    SGV3D_REQUIRE(d->mode == SGV3D_CONV_NORMAL && d->kh > 0 && d->kw > 0 && d->kh * d->kw <= (tapmajor ? 64 : 32) && d->stride > 0 &&
                      d->dil > 0 && d->pad >= 0,
                  "conv2d_x3_forward: NHWC output, at most 32 taps (64 with tap-major weights)");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0 && d->cin % 4 == 0 && d->cout % 4 == 0,
                  "conv2d_x3_forward: cin %% 4 == 0, cout %% 4 == 0 (cin=%d cout=%d)", d->cin, d->cout);
    SGV3D_REQUIRE(d->out_h == (d->in_h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1 &&
                      d->out_w == (d->in_w + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1,
                  "conv2d_x3_forward: output size does not match the geometry");
    SGV3D_REQUIRE((d->x_ld & 3) == 0 && (d->x_coff & 3) == 0 && (d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && (residual == nullptr || (d->res_ld & 3) == 0),
                  "conv2d_x3_forward: leading dimensions and channel offsets must be multiples of 4");
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout && (residual == nullptr || d->res_ld >= d->cout),
                  "conv2d_x3_forward: leading dimension too small");
    SGV3D_REQUIRE(d->cout_pad >= d->cout && d->cout_pad % 32 == 0, "conv2d_x3_forward: desc.cout_pad = rows of the x3 weights (a multiple of 32)");
    SGV3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) |
                    reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(u3_packed)) & 15) == 0,
                  "conv2d_x3_forward: pointers must be 16-B aligned");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    const long long K = tapmajor ? ((long long)d->kh * d->kw * d->cin + 31) / 32 * 32 : (long long)d->kh * d->kw * d->cin;
    const long long xb = (long long)d->batch * d->in_h * d->in_w * d->x_ld * 4, wb = (long long)d->cout_pad * K * 6;
    SGV3D_REQUIRE(M < 0x7fffffffLL && xb < 0xf0000000LL && wb < 0xf0000000LL, "conv2d_x3_forward: operands larger than 3.75 GiB");
    const int variant = d->tile & 15;
    SGV3D_REQUIRE((d->tile & SGV3D_TILE_X3) && (variant & 3) < 3 && (variant & 12) != 12,
                  "conv2d_x3_forward: desc.tile = SGV3D_TILE_X3 | variant, variant in {0,1,2, 4,5,6, 8,9,10}");
    const int taps = tapmajor ? 2 : (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0) ? 0 : 1;      // kernel MODE
    PwX3Args a;
    a.x = x; a.w = u3_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.M = (int)M; a.K = (int)K; a.N = d->cout;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.res_ld = d->res_ld; a.relu = d->relu;
    a.cout_pad = d->cout_pad;
    a.mfirst = (d->tile & SGV3D_TILE_MFIRST) ? 1 : 0;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.cin4 = d->cin / 4;
    const int bm = 32 << (variant & 3), bn = (variant & 8) ? 256 : (variant & 4) ? 64 : 128;
    a.tiles_m = (int)cdiv(M, bm);
    a.tiles_n = cdiv(d->cout, bn);
    hipStream_t st = as_stream(stream);
    a.split_k = d->split_k > 1 ? d->split_k : 1;
    a.ws = nullptr;
    if (a.split_k > 1) {
        SGV3D_REQUIRE(a.split_k <= K / 32, "conv2d_x3_forward: split_k %d exceeds the %lld k-steps", a.split_k, K / 32);
        const size_t need = (size_t)a.split_k * (size_t)M * d->cout * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(SGV3D_ENOSPACE, "conv2d_x3_forward: split-K workspace has %zu bytes, needs %zu", workspace_bytes, need);
        SGV3D_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "conv2d_x3_forward: workspace must be 16-B aligned");
        a.ws = static_cast<float *>(workspace);
    }
    int rc;
    switch (variant) {
        case 0: rc = launch_pw<2, 4>(a, taps, st); break;
        case 1: rc = launch_pw<4, 4>(a, taps, st); break;
        case 2: rc = launch_pw<8, 4>(a, taps, st); break;
        case 4: rc = launch_pw<2, 2>(a, taps, st); break;
        case 5: rc = launch_pw<4, 2>(a, taps, st); break;
        case 6: rc = launch_pw<8, 2>(a, taps, st); break;
        case 8: rc = launch_pw<2, 8>(a, taps, st); break;
        case 9: rc = launch_pw<4, 8>(a, taps, st); break;
        default: rc = launch_pw<8, 8>(a, taps, st); break;
    }
    if (rc != SGV3D_OK || a.split_k <= 1) return rc;
    // second stage: fixed-order sum of the partials + folded BN / bias, residual, ReLU (conv_igemm.hip)
    ConvArgs r{};
    r.M = a.M; r.N = a.N; r.ws = a.ws; r.split_k = a.split_k;
    r.scale = scale; r.bias = bias; r.res = residual; r.res_ld = d->res_ld; r.relu = d->relu;
    r.y = y; r.y_ld = d->y_ld; r.y_coff = d->y_coff; r.mode = SGV3D_CONV_NORMAL; r.cout = d->cout;
    r.m_h = d->out_h; r.m_w = d->out_w; r.out_h = d->out_h; r.out_w = d->out_w;
    return launch_splitk_reduce(r, st);
}
