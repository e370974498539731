// Weight gradient of the NHWC convolution on the fp32 MFMA pipe (the training step's
// replacement for the cuDNN backward-filter call behind nn.Conv2d / nn.ConvTranspose2d), plus the
// zero-insertion helper the data gradient of strided convolutions uses.
//
//   dW[co][ci][kh][kw] = sum over (image, oy, ox) of dY[image, oy, ox, co] * X[image, oy*s - p + kh*d, ox*s - p + kw*d, ci]
//
// GEMM view per kernel tap: C[co][ci] = sum_k A[k][co] * B[k][ci] with k = output pixel -- both operands are
// pixel-major in NHWC, which is exactly the operand shape of v_mfma_f32_32x32x2_f32 (lane l supplies
// A[k = l/32][m = l%32]), so tiles go global -> registers -> LDS [pixel][channel] without a transpose and the
// fragment reads are conflict-free ds_read_b32 of 32 consecutive floats per lane half.
//
// Workgroup = 256 threads = 2 x 2 waves, tile (64 WM) co x (64 WN) ci of ONE tap with WM, WN in {1, 2} (chosen per
// layer from its channel counts), 32 or 64 pixels per stage, loads issued two stages ahead into registers, LDS
// double-buffered with one barrier per stage.  The pixel range is split over blockIdx.y
// (the reduction is the long axis here: 5 000 - 83 000 pixels against 64 x 64 outputs); with more than one split
// the partial tiles go to the workspace and wgrad_reduce_kernel adds them in split order (deterministic) while
// transposing to the OIHW layout of nn.Conv2d.weight.grad.
#include "common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


struct WgradArgs {
    const float *x, *dy;
    float *dw, *ws;
    int batch, in_h, in_w, cin, out_h, out_w, cout, kh, kw, stride, pad, dil;
    int x_ld, x_coff, y_ld, y_coff;
    int tiles_co, tiles_ci, taps, split, pix_total, pix_per_split;
    int wm, wn;   // wave tiles of the chosen instantiation
    int tap_cols; // > 0: the ci axis of the tiles is the flat (tap, channel) list with tap_cols = cin rounded up to 4 columns per tap
                  // (the image stem: 16 taps x 4 channels per 64 columns; cin = 80 at 7x7: 62 tiles instead of 49 x 2 half-empty ones)
    unsigned x_bytes, y_bytes;   // extents of x / dy from their base pointers (buffer resources)
    // conv_wgrad3x3_kernel, batched form (blockIdx.z = problem): several dY / dW pairs against ONE x (the first layers of the
    // CenterHead branches all read the shared map); nbatch == 0: the single pair dy / dw
    int nbatch;
    const float *dy_list[48];
    float *dw_list[48];
};

typedef float f32x4n __attribute__((ext_vector_type(4)));

// WM x WN = 32x32 MFMA tiles per wave (2 x 2 waves: workgroup tile 64*WM co x 64*WN ci); STAGE = pixels per stage, chosen
// so that both LDS buffers of both operands take 64 KB (two workgroups per CU).  On this chip VALU instructions do not
// overlap with fp32 MFMAs (they share the lanes), so the staging arithmetic is pure overhead: the larger tiles halve the
// address / load / LDS-store instructions per MFMA.
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int STAGE = 8192 / (BM + BN) >= 64 ? 64 : 32;
    constexpr int A_ROWS = 1024 / BM, B_ROWS = 1024 / BN;       // pixels staged per pass of the 256 threads
    constexpr int A_PASS = STAGE / A_ROWS, B_PASS = STAGE / B_ROWS;
    __shared__ __attribute__((aligned(16))) float sA[2][STAGE][BM];
    __shared__ __attribute__((aligned(16))) float sB[2][STAGE][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
    int bid = blockIdx.x;
    const int tci = bid % a.tiles_ci; bid /= a.tiles_ci;
    const int tco = bid % a.tiles_co; bid /= a.tiles_co;
    // tap_cols: blockIdx enumerates groups of BN / 4 taps instead of (ci tile, tap); a thread's float4 of staged columns is
    // the 4 channels of ONE tap, so its tap (and whether it exists) is fixed for the whole kernel
    const int gcol4 = tci * BN + (int)(threadIdx.x % (BN / 4)) * 4;          // tap_cols: first of this thread's 4 staged columns
    const int tap = a.tap_cols ? gcol4 / a.tap_cols : bid;
    const bool tap_ok = tap < a.taps;
    const int th = tap / a.kw, tw = tap - th * a.kw;
    const int co0 = tco * BM, ci0 = a.tap_cols ? 0 : tci * BN;
    const int pix_begin = blockIdx.y * a.pix_per_split;
    const int pix_end = min(pix_begin + a.pix_per_split, a.pix_total);
    const int ac4 = (tid % (BM / 4)) * 4, arow = tid / (BM / 4);   // A: pixels arow + A_ROWS i, channels ac4 .. ac4 + 3
    const int bc4 = (tid % (BN / 4)) * 4, brow = tid / (BN / 4);
    const int hw = a.out_h * a.out_w;

    // Branch-free staging: buffer loads return zeros for the offset 0xffffffff (pixels past the range, taps that
    // fall outside the image).  Channel tails are loaded as they come: they only reach discarded rows / columns.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)a.y_bytes, 0x00020000);
    const unsigned x_c = (unsigned)(a.x_coff + (a.tap_cols ? gcol4 - tap * a.tap_cols : ci0 + bc4)) * 4u, y_c = (unsigned)(a.y_coff + co0 + ac4) * 4u;
    // (image, oy, ox) of the first pixel of the stage being loaded, advanced by STAGE pixels per stage
    int s_img = pix_begin / hw;
    int s_oy = (pix_begin - s_img * hw) / a.out_w;
    int s_ox = pix_begin - s_img * hw - s_oy * a.out_w;

    // Two register sets: the loads of stage s + 2 are issued before the MFMAs of stage s and land in LDS one stage later.
    f32x4n ra0[A_PASS], rb0[B_PASS], ra1[A_PASS], rb1[B_PASS];
    auto load_stage = [&](int p0, f32x4n (&ra)[A_PASS], f32x4n (&rb)[B_PASS]) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int pix = p0 + arow + A_ROWS * i;
            const unsigned yo = pix < pix_end ? (unsigned)pix * (unsigned)(a.y_ld * 4) + y_c : 0xffffffffu;
            ra[i] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, yo, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int off = brow + B_ROWS * i;
            int img = s_img, oy = s_oy, ox = s_ox + off;
            while (ox >= a.out_w) {
                ox -= a.out_w;
                if (++oy == a.out_h) { oy = 0; ++img; }
            }
            const int iy = oy * a.stride - a.pad + th * a.dil, ix = ox * a.stride - a.pad + tw * a.dil;
            const bool in_img = tap_ok && p0 + off < pix_end && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
            const unsigned xo = in_img ? (unsigned)((img * a.in_h + iy) * a.in_w + ix) * (unsigned)(a.x_ld * 4) + x_c : 0xffffffffu;
            rb[i] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xo, 0, 0));
        }
        s_ox += STAGE;
        while (s_ox >= a.out_w) {
            s_ox -= a.out_w;
            if (++s_oy == a.out_h) { s_oy = 0; ++s_img; }
        }
    };
    auto store_stage = [&](int buf, const f32x4n (&ra)[A_PASS], const f32x4n (&rb)[B_PASS]) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<f32x4n *>(&sA[buf][arow + A_ROWS * i][ac4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<f32x4n *>(&sB[buf][brow + B_ROWS * i][bc4]) = rb[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    // The STAGE / 2 k-steps of the stage in LDS buffer `buf`: fragments in batches of KB k-steps, the next batch in
    // flight under the MFMAs of the current one.
    constexpr int KB = WM * WN == 4 ? 4 : 8;     // k-steps per fragment batch (register budget of the 128 x 128 tile)
    constexpr int NB = STAGE / 2 / KB;
    auto compute = [&](int buf) {
        const float *pa = &sA[buf][half][wm * (BM / 2) + l32];
        const float *pb = &sB[buf][half][wn * (BN / 2) + l32];
        float fa[2][KB][WM], fb[2][KB][WN];
        auto read_batch = [&](int q, int set) {
#pragma unroll
            for (int j = 0; j < KB; ++j) {
#pragma unroll
                for (int m = 0; m < WM; ++m) fa[set][j][m] = pa[(q * KB + j) * 2 * BM + m * 32];
#pragma unroll
                for (int n = 0; n < WN; ++n) fb[set][j][n] = pb[(q * KB + j) * 2 * BN + n * 32];
            }
        };
        read_batch(0, 0);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (q + 1 < NB) read_batch(q + 1, (q + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < KB; ++j)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][j][m], fb[q & 1][j][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nst = (pix_end - pix_begin + STAGE - 1) / STAGE;
    load_stage(pix_begin, ra0, rb0);
    if (nst > 1) load_stage(pix_begin + STAGE, ra1, rb1);
    store_stage(0, ra0, rb0);
    __syncthreads();
    for (int st = 0; st < nst; st += 2) {
        if (st + 2 < nst) load_stage(pix_begin + (st + 2) * STAGE, ra0, rb0);
        compute(0);
        if (st + 1 < nst) store_stage(1, ra1, rb1);
        __syncthreads();
        if (st + 1 >= nst) break;
        if (st + 3 < nst) load_stage(pix_begin + (st + 3) * STAGE, ra1, rb1);
        compute(1);
        if (st + 2 < nst) store_stage(0, ra0, rb0);
        __syncthreads();
    }

#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int col = wn * (BN / 2) + n * 32 + l32;                 // column of the workgroup tile
            const int gcol = tci * BN + col;
            const int otap = a.tap_cols ? gcol / a.tap_cols : tap;
            const int ci = a.tap_cols ? gcol - otap * a.tap_cols : ci0 + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * (BM / 2) + m * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
                if (co >= a.cout || ci >= a.cin || otap >= a.taps) continue;
                if (a.split > 1)
                    a.ws[(((size_t)blockIdx.y * a.taps + otap) * a.cout + co) * a.cin + ci] = acc[m][n][e];
                else
                    a.dw[((size_t)co * a.cin + ci) * a.taps + otap] = acc[m][n][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / dilation 1: ALL NINE TAPS in one workgroup (tile id 5).  The per-tap kernel above makes nine workgroups
// load the same dY tile and nine shifted copies of the same X rows, and pays its staging address arithmetic (vector
// instructions = fp32 MFMA time on this chip) once per 32 MFMAs of a wave.  Here a stage is a SEGMENT OF ONE OUTPUT ROW (32
// pixels): dY [32][64 co] and the three input rows it touches, X [3][34][64 ci], go to LDS once and feed the nine taps'
// accumulators -- a wave owns a 32 co x 32 ci quadrant of the 64 x 64 tile for all taps (9 x 16 accumulator registers), a
// k-step is 1 dY fragment + 9 X fragments (the tap is an LDS address offset) + 9 MFMAs, a stage 144 MFMAs per wave against
// ~40 vector instructions of staging (row / column validity is per stage: two scalars per row, one compare per column).
// Work units = (image, output row, segment) split over blockIdx.y; partial tiles go to the workspace in the per-tap kernel's
// layout [split][tap][co][ci] and wgrad_reduce_kernel adds them in split order.
constexpr int kSeg = 32;                      // output pixels per stage
constexpr int kXCols = kSeg + 2;              // input columns per stage
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sY[2][kSeg][64];
    __shared__ __attribute__((aligned(16))) float sX[2][3][kXCols][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int segs = (a.out_w + kSeg - 1) / kSeg;
    const int units = a.batch * a.out_h * segs;
    const int u_begin = (int)((long long)units * blockIdx.y / a.split), u_end = (int)((long long)units * (blockIdx.y + 1) / a.split);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const float *const dyp = a.nbatch ? a.dy_list[blockIdx.z] : a.dy;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dyp, 0, (int)a.y_bytes, 0x00020000);
    // staging slots: a thread moves float4s.  dY: 32 pixels x 16 float4 = 2 per thread; X: 3 rows x 34 columns x 16 float4 =
    // 1632 = 6.4 per thread (7 passes, the last one partly idle).  Row / column / channel of a slot never change.
    const int c4 = (tid & 15) * 4;
    constexpr int XP = (3 * kXCols * 16 + 255) / 256;
    int x_row[XP], x_col[XP];
    unsigned x_off[XP];                            // byte offset relative to the unit's base pixel (row 0, column 0 of the patch)
    bool x_any[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int slot = (tid >> 4) + 16 * i;     // (row, column) index, 0 .. 3 * 34 - 1
        x_any[i] = slot < 3 * kXCols;
        x_row[i] = slot / kXCols;
        x_col[i] = slot - x_row[i] * kXCols;
        x_off[i] = (unsigned)((x_row[i] * a.in_w + x_col[i]) * a.x_ld * 4 + (a.x_coff + ci0 + c4) * 4);
    }
    const bool ci_ok = ci0 + c4 < a.cin, co_ok = co0 + c4 < a.cout;     // (channel tails: whole float4s, cin / cout % 4 == 0)
    const int y_p0 = tid >> 4;                     // dY pixels y_p0, y_p0 + 16
    f32x4n rx[XP], ry[2];
    auto load_unit = [&](int u) {
        const int seg = u % segs, t = u / segs;
        const int oy = t % a.out_h, img = t / a.out_h;
        const int ox0 = seg * kSeg;
        const int iy0 = oy - a.pad, ix0 = ox0 - a.pad;                     // top-left input pixel of the patch
        const long long base = ((long long)(img * a.in_h + iy0) * a.in_w + ix0) * a.x_ld * 4;   // may be negative: wraps, only used when valid
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const bool ok = x_any[i] && ci_ok && u < u_end && (unsigned)(iy0 + x_row[i]) < (unsigned)a.in_h &&
                            (unsigned)(ix0 + x_col[i]) < (unsigned)a.in_w;
            rx[i] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? (unsigned)base + x_off[i] : 0xffffffffu, 0, 0));
        }
        const unsigned ybase = (unsigned)(((long long)(img * a.out_h + oy) * a.out_w + ox0) * a.y_ld * 4 + (a.y_coff + co0 + c4) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = y_p0 + 16 * i;
            const bool ok = co_ok && u < u_end && ox0 + p < a.out_w;
            ry[i] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, ok ? ybase + (unsigned)(p * a.y_ld * 4) : 0xffffffffu, 0, 0));
        }
    };
    auto store_unit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XP; ++i)
            if (x_any[i]) *reinterpret_cast<f32x4n *>(&sX[buf][x_row[i]][x_col[i]][c4]) = rx[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4n *>(&sY[buf][y_p0 + 16 * i][c4]) = ry[i];
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    auto compute = [&](int buf) {
        const float *py = &sY[buf][half][wm * 32 + l32];
        const float *px = &sX[buf][0][half][wn * 32 + l32];
        float fa[2], fb[2][9];
        auto read_step = [&](int j, int set) {
            fa[set] = py[j * 2 * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) fb[set][t] = px[((t / 3) * kXCols + j * 2 + (t % 3)) * 64];
        };
        read_step(0, 0);
#pragma unroll
        for (int j = 0; j < kSeg / 2; ++j) {
            if (j + 1 < kSeg / 2) read_step(j + 1, (j + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1], fb[j & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (u_begin < u_end) {
        load_unit(u_begin);
        store_unit(0);
        __syncthreads();
        for (int u = u_begin; u < u_end; ++u) {
            const int buf = (u - u_begin) & 1;
            load_unit(u + 1);                          // (past the end: every request out of range)
            compute(buf);
            store_unit(buf ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ci = ci0 + wn * 32 + l32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
            if (co >= a.cout || ci >= a.cin) continue;
            if (a.split > 1)
                a.ws[((((size_t)blockIdx.z * a.split + blockIdx.y) * 9 + t) * a.cout + co) * a.cin + ci] = acc[t][e];
            else
                (a.nbatch ? a.dw_list[blockIdx.z] : a.dw)[((size_t)co * a.cin + ci) * 9 + t] = acc[t][e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / dilation 1 with 1 .. 4 OUTPUT channels (the 36 final layers of the CenterHead branches: 64 -> 1 / 2 / 3): an
// MFMA tile would be 97 % padding (the per-tap kernel runs these at 2 TFLOP/s, 160 us each, 5.7 ms of a training step).  Here a
// lane owns one input channel, a wave walks output-row segments of 32 pixels, 16 at a time: the 3 x 18 input values of its channel
// in registers (coalesced 256-byte rows, 48 loads in flight), the dY values of 16 pixels come with ONE coalesced load and
// are broadcast with v_readlane, and the 9 x COUT products per pixel are plain FMAs into 9 x COUT accumulators per lane.
// The four waves of a workgroup are added in LDS in wave order, partial sums per workgroup go to the workspace
// [workgroup][tap][c][ci] and wgrad_thin_reduce_kernel adds them in workgroup order (deterministic).
// Bound: vector ALU (9 COUT FMAs per pixel and channel) / HBM (X is read once per launch: 33.5 MB for a cfg-2 hidden map).
struct ThinArgs {
    const float *x, *dy;
    float *ws, *dw;
    int batch, in_h, in_w, out_h, out_w, pad, cin, cout, x_ld, x_coff, y_ld, y_coff;
    int segs, units, waves;        // 128-pixel segments per output row, units = batch * out_h * segs, waves = gridDim.x * 4
    unsigned x_bytes, y_bytes;
};
constexpr int kThinSeg = 32;

template <int COUT>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const ThinArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;                     // global wave index
    const int ci = blockIdx.y * 64 + lane;
    const bool ci_ok = ci < a.cin;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)a.y_bytes, 0x00020000);
    float acc[9][COUT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[t][c] = 0.f;
    const unsigned x_c = (unsigned)(a.x_coff + ci) * 4u;
    for (int u = gw; u < a.units; u += a.waves) {
        const int seg = u % a.segs, t0 = u / a.segs;
        const int oy = t0 % a.out_h, img = t0 / a.out_h;
        const int ox_begin = seg * kThinSeg, ox_end = min(ox_begin + kThinSeg, a.out_w);
        const int iy0 = oy - a.pad;
        // byte offsets of the three input rows at column 0 (out of range: the row is outside the image)
        unsigned rowoff[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = iy0 + r;
            rowoff[r] = (ci_ok && (unsigned)iy < (unsigned)a.in_h) ? (unsigned)((img * a.in_h + iy) * a.in_w) * (unsigned)(a.x_ld * 4) + x_c : 0xffffffffu;
        }
        auto load_x = [&](int r, int ix) -> float {
            const unsigned off = (rowoff[r] != 0xffffffffu && (unsigned)ix < (unsigned)a.in_w) ? rowoff[r] + (unsigned)ix * (unsigned)(a.x_ld * 4) : 0xffffffffu;
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, 0, 0));
        };
        // 16 pixels at a time: their 3 x 18 input values are requested together (48 loads in flight per lane, two columns carried
        // over), then 16 x 9 x COUT FMAs
        float w[3][18];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            w[r][0] = load_x(r, ox_begin - a.pad);
            w[r][1] = load_x(r, ox_begin - a.pad + 1);
        }
        const unsigned ybase = (unsigned)((img * a.out_h + oy) * a.out_w) * (unsigned)(a.y_ld * 4) + (unsigned)a.y_coff * 4u;
        for (int ox0 = ox_begin; ox0 < ox_end; ox0 += 16) {
            // dY of 16 pixels: lane l holds element l of the row's flat (pixel, channel) array starting at pixel ox0
            const int px = lane / a.y_ld;                      // (y_ld <= 4: at least 16 pixels in 64 lanes)
            const bool yok = ox0 + px < ox_end && px < 16;     // pixels past the row's end multiply dY = 0
            const float dyv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                y_rsrc, yok ? ybase + (unsigned)(ox0 * a.y_ld + lane) * 4u : 0xffffffffu, 0, 0));
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) w[r][2 + j] = load_x(r, ox0 - a.pad + 2 + j);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int c = 0; c < COUT; ++c) {
                    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyv), j * a.y_ld + c));
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < 3; ++s2) acc[r * 3 + s2][c] = __builtin_fmaf(d, w[r][j + s2], acc[r * 3 + s2][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) { w[r][0] = w[r][16]; w[r][1] = w[r][17]; }
        }
    }
    // the four waves of the workgroup meet in LDS and are added in wave order; one partial set per workgroup
    __shared__ float red[4][9 * COUT][64];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) red[wave][t * COUT + c][lane] = acc[t][c];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * COUT * 64; e += 256) {
        const int l = e & 63, tc = e >> 6;
        const int cc = blockIdx.y * 64 + l;
        if (cc < a.cin) a.ws[((size_t)blockIdx.x * 9 * COUT + tc) * a.cin + cc] = ((red[0][tc][l] + red[1][tc][l]) + red[2][tc][l]) + red[3][tc][l];
    }
}

// dw[c][ci][tap] = sum over workgroups (in order) of ws[workgroup][tap][c][ci]
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const ThinArgs a) {
    const int total = 9 * a.cout * a.cin;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    const int parts = a.waves >> 2;                            // one partial set per workgroup
    int p = 0;
    for (; p + 8 <= parts; p += 8) {                           // eight independent loads in flight, added in order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = a.ws[(size_t)(p + j) * total + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    for (; p < parts; ++p) v += a.ws[(size_t)p * total + i];
    const int ci = i % a.cin, r = i / a.cin;
    const int c = r % a.cout, tap = r / a.cout;
    a.dw[((size_t)c * a.cin + ci) * 9 + tap] = v;
}

// i = (tap, co, ci) with ci fastest: coalesced partial reads, split order fixed.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a) {
    const long long total = (long long)a.taps * a.cout * a.cin;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    const float *ws = a.ws + (size_t)blockIdx.y * a.split * total;           // (batched form: one slab of partials per problem)
    for (int s = 0; s < a.split; ++s) v += ws[(size_t)s * total + i];
    const int ci = (int)(i % a.cin);
    const long long r = i / a.cin;
    const int co = (int)(r % a.cout), tap = (int)(r / a.cout);
    (a.nbatch ? a.dw_list[blockIdx.y] : a.dw)[((size_t)co * a.cin + ci) * a.taps + tap] = v;
}

__global__ __launch_bounds__(256) void zero_insert_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, int batch,
                                                           int in_h, int in_w, int c4, int stride, int out_h, int out_w) {
    const long long total = (long long)batch * out_h * out_w * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4);
        long long r = i / c4;
        const int ox = (int)(r % out_w); r /= out_w;
        const int oy = (int)(r % out_h);
        const int img = (int)(r / out_h);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy % stride == 0 && ox % stride == 0 && oy / stride < in_h && ox / stride < in_w)
            v = x[((size_t)(img * in_h + oy / stride) * in_w + ox / stride) * c4 + c];
        y[i] = v;
    }
}

struct PhaseArgs {
    const float4 *src[4];     // phase (py, px) = src[py * 2 + px]: NHWC [batch, h[p], w[p], c4 * 4]
    int h[4], w[4], r0[4], c0[4];
    float4 *dst;              // NHWC [batch, out_h, out_w, c4 * 4]
    int batch, out_h, out_w, c4;
};

// dst[b, 2 i + py, 2 j + px, :] = src[py][px][b, i + r0, j + c0, :]  (the four sub-pixel phases of a stride-2 data gradient)
__global__ __launch_bounds__(256) void interleave_phases_kernel(const PhaseArgs a) {
    const long long total = (long long)a.batch * a.out_h * a.out_w * a.c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.c4);
        long long r = i / a.c4;
        const int x = (int)(r % a.out_w); r /= a.out_w;
        const int y = (int)(r % a.out_h);
        const int b = (int)(r / a.out_h);
        const int p = (y & 1) * 2 + (x & 1);
        const int sy = (y >> 1) + a.r0[p], sx = (x >> 1) + a.c0[p];
        a.dst[i] = a.src[p][((size_t)(b * a.h[p] + sy) * a.w[p] + sx) * a.c4 + c];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 matrix cores (mixed-precision training, BASELINE configs[4]: "ResNet-101 bf16"; the reference trains with
// --amp_backend native, docs/run_and_eval.md:5,16): the per-tap GEMM C[co][ci] = sum_p dY[p][co] X[p + tap][ci] on
// v_mfma_f32_32x32x16_bf16 -- f32 tensors in HBM (master copies, gradients and activations stay f32), operands rounded to
// bf16 (nearest even) on their way into LDS, f32 accumulation, the same partial-tile workspace and fixed-order reduce as the
// f32 kernel.
//
// The reduction index (the pixel) is the SLOW axis of both operands in NHWC, and the MFMA wants 8 consecutive k per lane: the
// transpose happens in registers while staging.  A thread owns a block of 8 pixels x 4 channels (eight 16-byte loads, one
// pixel each), rounds it and stores it as four 16-byte rows of 8 pixels into an LDS image laid out [channel][64 pixels] -- the
// operand layout: lane (m = l % 32, h = l / 32) reads pixels 16 ks + 8 h .. + 8 of channel row m with one ds_read_b128.
// The 16-byte chunk c of row r sits at slot c ^ key(r), key(r) = ((r >> 2) & 7) ^ (((r >> 1) & 1) << 2): the eight lanes of a
// store group (consecutive channel quads, one pixel group) and the sixteen lanes of a read group ({0-3, 12-15, 20-27} ...
// of 32 consecutive rows, one chunk) fall on distinct 16-byte slots.
//
// Workgroup = 256 threads = 2 x 2 waves, tile (64 TM) co x (64 TN) ci of one tap, 64 pixels per stage (4 k-steps of 16), one
// register stage + two LDS buffers.  128 x 128: 32 flop per byte loaded -- the f32 loads of 64 pixels x 256 channels per stage
// are what bounds it (L2), not the MFMA pipe.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x8 __attribute__((ext_vector_type(8)));

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradArgs a) {
    constexpr int BM = 64 * TM, BN = 64 * TN, STAGE = 64;
    constexpr int NA = 8 * (BM / 4), NB_ = 8 * (BN / 4);                    // (8 pixels x 4 channels) blocks per stage
    constexpr int B_OFF = (TM == 1 && TN == 1) ? 128 : 0;                   // 64 x 64: threads 0-127 stage dY, 128-255 stage X
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * STAGE];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
    int bid = blockIdx.x;
    const int tci = bid % a.tiles_ci; bid /= a.tiles_ci;
    const int tco = bid % a.tiles_co; bid /= a.tiles_co;
    const int tap = bid;
    const int th = tap / a.kw, tw = tap - th * a.kw;
    const int co0 = tco * BM, ci0 = tci * BN;
    const int pix_begin = blockIdx.y * a.pix_per_split;
    const int pix_end = min(pix_begin + a.pix_per_split, a.pix_total);
    const int hw = a.out_h * a.out_w;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    // (batched form, blockIdx.z = problem: several dY / dW pairs against ONE x, as conv_wgrad3x3_kernel)
    const float *const dyp = a.nbatch ? a.dy_list[blockIdx.z] : a.dy;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dyp, 0, (int)a.y_bytes, 0x00020000);

    // staging blocks of this thread
    const int ia = tid, ib = tid - B_OFF;
    const bool has_a = ia < NA, has_b = ib >= 0 && ib < NB_;
    const int a_pg = ia / (BM / 4), a_cq = ia % (BM / 4);
    const int b_pg = has_b ? ib / (BN / 4) : 0, b_cq = has_b ? ib % (BN / 4) : 0;
    // channel tails: whole float4s (cin / cout % 4 == 0 is required by the entry); a quad past the layer's channels reads zeros
    const bool a_ch_ok = has_a && co0 + 4 * a_cq < a.cout, b_ch_ok = has_b && ci0 + 4 * b_cq < a.cin;
    const unsigned y_c = (unsigned)(a.y_coff + co0 + 4 * a_cq) * 4u, x_c = (unsigned)(a.x_coff + ci0 + 4 * b_cq) * 4u;
    // LDS store addresses (bytes inside a buffer): rows 4 cq + e, chunk pg at slot pg ^ key(row)
    unsigned a_st[4], b_st[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a_st[e] = (unsigned)((4 * a_cq + e) * 128 + ((a_pg ^ ((a_cq & 7) ^ ((e >> 1) << 2))) << 4));
        b_st[e] = (unsigned)((4 * b_cq + e) * 128 + ((b_pg ^ ((b_cq & 7) ^ ((e >> 1) << 2))) << 4));
    }
    // fragment read offsets: row = wave offset + t * 32 + l32 (key depends on l32 only), chunk 2 ks + half
    const int rkey = ((l32 >> 2) & 7) ^ (((l32 >> 1) & 1) << 2);
    unsigned rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd[ks] = (unsigned)(l32 * 128 + (((2 * ks + half) ^ rkey) << 4));
    const unsigned a_row0 = (unsigned)(wm * (BM / 2) * 128), b_row0 = (unsigned)(wn * (BN / 2) * 128);

    f32x4n ra[8], rb[8];
    auto load_stage = [&](int p0) {
        // dY: pixels p0 + 8 a_pg + j
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pix = p0 + 8 * a_pg + j;
            const unsigned yo = (a_ch_ok && pix < pix_end) ? (unsigned)pix * (unsigned)(a.y_ld * 4) + y_c : 0xffffffffu;
            ra[j] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, yo, 0, 0));
        }
        // X: the tap's input pixel of output pixels p0 + 8 b_pg + j (one decode per block, then a walk along the row)
        const int q0 = p0 + 8 * b_pg;
        int img = q0 / hw;
        int r = q0 - img * hw;
        int oy = r / a.out_w;
        int ox = r - oy * a.out_w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = oy * a.stride - a.pad + th * a.dil, ix = ox * a.stride - a.pad + tw * a.dil;
            const bool ok = b_ch_ok && q0 + j < pix_end && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
            const unsigned xo = ok ? (unsigned)((img * a.in_h + iy) * a.in_w + ix) * (unsigned)(a.x_ld * 4) + x_c : 0xffffffffu;
            rb[j] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xo, 0, 0));
            if (++ox == a.out_w) { ox = 0; if (++oy == a.out_h) { oy = 0; ++img; } }
        }
    };
    auto store_stage = [&](int buf) {
        char *pa = reinterpret_cast<char *>(sA[buf]), *pb = reinterpret_cast<char *>(sB[buf]);
        if (has_a) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const wf32x8 v = {ra[0][e], ra[1][e], ra[2][e], ra[3][e], ra[4][e], ra[5][e], ra[6][e], ra[7][e]};
                *reinterpret_cast<wbf16x8 *>(pa + a_st[e]) = __builtin_convertvector(v, wbf16x8);
            }
        }
        if (has_b) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const wf32x8 v = {rb[0][e], rb[1][e], rb[2][e], rb[3][e], rb[4][e], rb[5][e], rb[6][e], rb[7][e]};
                *reinterpret_cast<wbf16x8 *>(pb + b_st[e]) = __builtin_convertvector(v, wbf16x8);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    auto compute = [&](int buf) {
        const char *pa = reinterpret_cast<const char *>(sA[buf]) + a_row0, *pb = reinterpret_cast<const char *>(sB[buf]) + b_row0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wbf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) fa[m] = *reinterpret_cast<const wbf16x8 *>(pa + m * 32 * 128 + rd[ks]);
#pragma unroll
            for (int n = 0; n < TN; ++n) fb[n] = *reinterpret_cast<const wbf16x8 *>(pb + n * 32 * 128 + rd[ks]);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
    };

    const int nst = (pix_end - pix_begin + STAGE - 1) / STAGE;
    load_stage(pix_begin);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) load_stage(pix_begin + (st + 1) * STAGE);     // in flight under this stage's MFMAs
        compute(buf);
        if (st + 1 < nst) store_stage(buf ^ 1);
        __syncthreads();
    }

    // (the accumulator layout and the output / workspace layout of conv_wgrad_kernel)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int ci = ci0 + wn * (BN / 2) + n * 32 + l32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * (BM / 2) + m * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
                if (co >= a.cout || ci >= a.cin) continue;
                if (a.split > 1)
                    a.ws[((((size_t)blockIdx.z * a.split + blockIdx.y) * a.taps + tap) * a.cout + co) * a.cin + ci] = acc[m][n][e];
                else
                    (a.nbatch ? a.dw_list[blockIdx.z] : a.dw)[((size_t)co * a.cin + ci) * a.taps + tap] = acc[m][n][e];
            }
        }
}

int fill_args(const sgv3d_conv_desc *d, int split, WgradArgs &a, int tile_override = 0) {
    SGV3D_REQUIRE(d, "conv2d_backward_weight: null descriptor");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0 && d->out_h > 0 && d->out_w > 0,
                  "conv2d_backward_weight: bad sizes");
    SGV3D_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv2d_backward_weight: bad kernel geometry");
    SGV3D_REQUIRE(d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout, "conv2d_backward_weight: channel strides too small");
    SGV3D_REQUIRE(d->out_h == (d->in_h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1 &&
                  d->out_w == (d->in_w + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1,
                  "conv2d_backward_weight: output size does not belong to this input size");
    const long long pix = (long long)d->batch * d->out_h * d->out_w;
    SGV3D_REQUIRE(pix < (1ll << 31) && (long long)d->batch * d->in_h * d->in_w < (1ll << 31), "conv2d_backward_weight: too many pixels");
    a = WgradArgs{};
    a.batch = d->batch; a.in_h = d->in_h; a.in_w = d->in_w; a.cin = d->cin; a.out_h = d->out_h; a.out_w = d->out_w;
    a.cout = d->cout; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    // tile: the padded MFMA work divided by the measured relative efficiency of the instantiation.  On the cfg-2 layers
    // the 64 x 64 tile wins almost everywhere (more workgroups per round outweigh the fewer staging instructions per
    // MFMA of the larger tiles), so the larger ones are only chosen when they pad less.
    double best = 0;
    for (int wm = 1; wm <= 2; ++wm)
        for (int wn = 1; wn <= 2; ++wn) {
            const double eff = wm * wn == 1 ? 1.0 : 0.85;
            const double cost = (double)cdiv(d->cout, 64 * wm) * (64 * wm) * cdiv(d->cin, 64 * wn) * (64 * wn) / eff;
            if (best == 0 || cost < best) { best = cost; a.wm = wm; a.wn = wn; }
        }
    const bool all_taps = tile_override == 5;     // conv_wgrad3x3_kernel: all nine taps of a 64 x 64 tile in one workgroup
    if (all_taps) {
        SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_coff % 4 == 0 &&
                          d->y_coff % 4 == 0 && d->x_ld % 4 == 0 && d->y_ld % 4 == 0,
                      "conv2d_backward_weight: tile 5 covers 3x3 / stride 1 / dilation 1 layers with channel counts and offsets that are multiples of 4");
        a.wm = a.wn = 1;
    } else if (tile_override > 0) { a.wm = ((tile_override - 1) >> 1) ? 2 : 1; a.wn = ((tile_override - 1) & 1) ? 2 : 1; }
    a.taps = d->kh * d->kw;
    // flat (tap, channel) columns when that needs fewer 64-column tiles than one tile row per tap: the 3-channel image stem (x_ld 4),
    // cin = 80 / 160 / 320 with 7x7 or strided 3x3 kernels
    {
        const int cpt = (d->cin + 3) / 4 * 4;
        const bool can = !all_taps && a.taps > 1 && d->x_ld % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + cpt <= d->x_ld;
        const long long flat = cdiv((long long)a.taps * cpt, 64), per_tap = (long long)a.taps * cdiv(d->cin, 64 * a.wn) * a.wn;
        a.tap_cols = (can && flat * 10 < per_tap * 9) ? cpt : 0;
    }
    if (a.tap_cols) a.wn = 1;
    const int stage_pix = 8192 / (64 * a.wm + 64 * a.wn) >= 64 ? 64 : 32;
    a.tiles_co = cdiv(d->cout, 64 * a.wm);
    a.tiles_ci = a.tap_cols ? cdiv((long long)a.taps * a.tap_cols, 64 * a.wn) : cdiv(d->cin, 64 * a.wn);
    a.pix_total = (int)pix;
    const long long tiles = (long long)a.tiles_co * a.tiles_ci * (a.tap_cols ? 1 : a.taps);
    SGV3D_REQUIRE(tiles < (1ll << 31), "conv2d_backward_weight: too many tiles");
    const int stages = cdiv(pix, stage_pix);
    if (split <= 0) {   // measured on cfg-2 layers: ~64 pixel ranges per tile, between 1 and 6 workgroups per CU in total
        long long target = tiles * 64;
        target = target < 256 ? 256 : (target > 1536 ? 1536 : target);
        split = (int)((target + tiles - 1) / tiles);
        split = split < 1 ? 1 : split;
        const int cap = stages / 4 > 0 ? stages / 4 : 1;
        split = split > cap ? cap : split;
    }
    if (all_taps) {     // work units = (image, output row, 32-pixel segment); a workgroup per (co tile, ci tile, unit range)
        const long long units = (long long)d->batch * d->out_h * cdiv(d->out_w, 32);
        const long long t2 = (long long)a.tiles_co * a.tiles_ci;
        if (split <= 0) {
            split = (int)((512 + t2 - 1) / t2);         // ~2 workgroups per CU in total
            const long long cap = units / 4 > 0 ? units / 4 : 1;
            split = split > cap ? (int)cap : split;
        }
        split = split < 1 ? 1 : split;
        split = split > units ? (int)units : split;
        split = split > 65535 ? 65535 : split;
        a.pix_per_split = 0;
        a.split = split;
    } else {
        split = split > stages ? stages : split;
        split = split > 65535 ? 65535 : split;
        a.pix_per_split = cdiv(stages, split) * stage_pix;
        a.split = cdiv(pix, a.pix_per_split);
    }
    const unsigned long long xb = (unsigned long long)d->batch * d->in_h * d->in_w * d->x_ld * 4ull;
    const unsigned long long yb = (unsigned long long)pix * d->y_ld * 4ull;
    SGV3D_REQUIRE(xb < 0xf0000000ull && yb < 0xf0000000ull, "conv2d_backward_weight: x / dy must be smaller than 3.75 GiB");
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
    return SGV3D_OK;
}

}  // namespace

extern "C" size_t sgv3d_conv2d_backward_weight_workspace_bytes(const sgv3d_conv_desc *d, int split) {
    WgradArgs a;
    if (fill_args(d, split, a, d ? d->tile : 0) != SGV3D_OK) return 0;
    return a.split > 1 ? (size_t)a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
}

extern "C" int sgv3d_conv2d_backward_weight(const sgv3d_conv_desc *d, const float *x, const float *dy, float *dw,
                                            int split, void *workspace, size_t workspace_bytes, void *stream) {
    WgradArgs a;
    if (int rc = fill_args(d, split, a, d ? d->tile : 0)) return rc;
    SGV3D_REQUIRE(x && dy && dw, "conv2d_backward_weight: null pointer");
    SGV3D_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)dy & 3) == 0, "conv2d_backward_weight: x / dy must be 4-byte aligned");
    const size_t need = a.split > 1 ? (size_t)a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
    SGV3D_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "conv2d_backward_weight: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    a.x = x; a.dy = dy; a.dw = dw; a.ws = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    const dim3 grid(a.tiles_co * a.tiles_ci * (a.tap_cols ? 1 : a.taps), a.split);
    if (d->tile == 5) conv_wgrad3x3_kernel<<<dim3(a.tiles_co * a.tiles_ci, a.split), 256, 0, st>>>(a);
    else if (a.wm == 2 && a.wn == 2) conv_wgrad_kernel<2, 2><<<grid, 256, 0, st>>>(a);
    else if (a.wm == 2) conv_wgrad_kernel<2, 1><<<grid, 256, 0, st>>>(a);
    else if (a.wn == 2) conv_wgrad_kernel<1, 2><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_kernel<1, 1><<<grid, 256, 0, st>>>(a);
    if (int rc = check_launch("conv_wgrad_kernel")) return rc;
    if (a.split > 1) {
        const long long total = (long long)a.taps * a.cout * a.cin;
        wgrad_reduce_kernel<<<cdiv(total, 256), 256, 0, st>>>(a);
        return check_launch("wgrad_reduce_kernel");
    }
    return SGV3D_OK;
}

// The same gradient with the products on the bf16 matrix cores (conv_wgrad_bf16_kernel: f32 tensors, operands rounded to bf16
// while staging, f32 accumulation).  desc.tile: 0 = 128 x 128 where both channel counts exceed 64, else 64 x 64; 1 = 64 x 64;
// 4 = 128 x 128.  Needs channel counts, strides and offsets that are multiples of 4 and 16-byte aligned tensors.  Workspace and
// split as sgv3d_conv2d_backward_weight (sgv3d_conv2d_backward_weight_bf16_workspace_bytes).
namespace {
int fill_args_bf16(const sgv3d_conv_desc *d, int split, WgradArgs &a) {
    SGV3D_REQUIRE(d, "conv2d_backward_weight_bf16: null descriptor");
    const int t = d->tile == 0 ? ((d->cout > 64 && d->cin > 64) ? 4 : 1) : d->tile;
    SGV3D_REQUIRE(t == 1 || t == 4, "conv2d_backward_weight_bf16: tile must be 0, 1 (64 x 64) or 4 (128 x 128), got %d", d->tile);
    SGV3D_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_coff % 4 == 0 && d->y_coff % 4 == 0 && d->x_ld % 4 == 0 && d->y_ld % 4 == 0,
                  "conv2d_backward_weight_bf16: channel counts, strides and offsets must be multiples of 4");
    if (int rc = fill_args(d, split, a, t)) return rc;
    if (a.tap_cols) {          // (the flat tap-channel columns of the image stem are an f32-kernel layout: per-tap tiles here)
        a.tap_cols = 0;
        a.wn = (t == 4) ? 2 : 1;
        a.tiles_ci = cdiv(d->cin, 64 * a.wn);
    }
    return SGV3D_OK;
}
}  // namespace

extern "C" size_t sgv3d_conv2d_backward_weight_bf16_workspace_bytes(const sgv3d_conv_desc *d, int split) {
    WgradArgs a;
    if (fill_args_bf16(d, split, a) != SGV3D_OK) return 0;
    return a.split > 1 ? (size_t)a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
}

extern "C" int sgv3d_conv2d_backward_weight_bf16(const sgv3d_conv_desc *d, const float *x, const float *dy, float *dw, int split,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
    WgradArgs a;
    if (int rc = fill_args_bf16(d, split, a)) return rc;
    SGV3D_REQUIRE(x && dy && dw, "conv2d_backward_weight_bf16: null pointer");
    SGV3D_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv2d_backward_weight_bf16: x / dy must be 16-byte aligned");
    const size_t need = a.split > 1 ? (size_t)a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
    SGV3D_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "conv2d_backward_weight_bf16: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    a.x = x; a.dy = dy; a.dw = dw; a.ws = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    const dim3 grid(a.tiles_co * a.tiles_ci * a.taps, a.split);
    if (a.wm == 2 && a.wn == 2) conv_wgrad_bf16_kernel<2, 2><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_bf16_kernel<1, 1><<<grid, 256, 0, st>>>(a);
    if (int rc = check_launch("conv_wgrad_bf16_kernel")) return rc;
    if (a.split > 1) {
        const long long total = (long long)a.taps * a.cout * a.cin;
        wgrad_reduce_kernel<<<cdiv(total, 256), 256, 0, st>>>(a);
        return check_launch("wgrad_reduce_kernel");
    }
    return SGV3D_OK;
}

// Batched form: n weight gradients dw_list[i] = wgrad(x, dy_list[i]) of n layers that read the SAME input (desc describes one of
// them), one launch (blockIdx.z = problem) -- the 36 first layers of the CenterHead branches in the mixed-precision step.
extern "C" size_t sgv3d_conv2d_backward_weight_bf16_batched_workspace_bytes(const sgv3d_conv_desc *d, int n, int split) {
    WgradArgs a;
    if (!d || n <= 0 || fill_args_bf16(d, split, a) != SGV3D_OK) return 0;
    return a.split > 1 ? (size_t)n * a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
}

extern "C" int sgv3d_conv2d_backward_weight_bf16_batched(const sgv3d_conv_desc *d, const float *x, const float *const *dy_list,
                                                         float *const *dw_list, int n, int split, void *workspace,
                                                         size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(d && x && dy_list && dw_list && n > 0 && n <= 48, "conv2d_backward_weight_bf16_batched: 1 .. 48 problems");
    WgradArgs a;
    if (int rc = fill_args_bf16(d, split, a)) return rc;
    SGV3D_REQUIRE(((uintptr_t)x & 15) == 0, "conv2d_backward_weight_bf16_batched: x must be 16-byte aligned");
    const size_t need = a.split > 1 ? (size_t)n * a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
    SGV3D_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "conv2d_backward_weight_bf16_batched: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    a.x = x; a.dy = nullptr; a.dw = nullptr; a.ws = static_cast<float *>(workspace);
    a.nbatch = n;
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(dy_list[i] && dw_list[i] && ((uintptr_t)dy_list[i] & 15) == 0, "conv2d_backward_weight_bf16_batched: null / unaligned pointer %d", i);
        a.dy_list[i] = dy_list[i];
        a.dw_list[i] = dw_list[i];
    }
    hipStream_t st = as_stream(stream);
    const dim3 grid(a.tiles_co * a.tiles_ci * a.taps, a.split, n);
    if (a.wm == 2 && a.wn == 2) conv_wgrad_bf16_kernel<2, 2><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_bf16_kernel<1, 1><<<grid, 256, 0, st>>>(a);
    if (int rc = check_launch("conv_wgrad_bf16_kernel(batched)")) return rc;
    if (a.split > 1) {
        const long long total = (long long)a.taps * a.cout * a.cin;
        wgrad_reduce_kernel<<<dim3(cdiv(total, 256), n), 256, 0, st>>>(a);
        return check_launch("wgrad_reduce_kernel");
    }
    return SGV3D_OK;
}

// Batched form of the all-taps kernel: n weight gradients dw_list[i] = wgrad(x, dy_list[i]) of n 3x3 / stride-1 layers that read the
// SAME input (desc describes one of them) in one launch -- the 36 first layers of the CenterHead branches (64 -> 64 at 256 x 256) are
// one 64 x 64 tile each: alone a launch needs a split of 256 to fill the chip and writes 256 partial tiles, batched 36 x 14.
extern "C" size_t sgv3d_conv2d_backward_weight_batched_workspace_bytes(const sgv3d_conv_desc *d, int n, int split) {
    WgradArgs a;
    if (!d || n <= 0 || fill_args(d, split, a, 5) != SGV3D_OK) return 0;
    return a.split > 1 ? (size_t)n * a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
}

extern "C" int sgv3d_conv2d_backward_weight_batched(const sgv3d_conv_desc *d, const float *x, const float *const *dy_list,
                                                    float *const *dw_list, int n, int split, void *workspace, size_t workspace_bytes,
                                                    void *stream) {
    SGV3D_REQUIRE(d && x && dy_list && dw_list && n > 0 && n <= 48, "conv2d_backward_weight_batched: 1 .. 48 problems");
    WgradArgs a;
    if (int rc = fill_args(d, split, a, 5)) return rc;
    const size_t need = a.split > 1 ? (size_t)n * a.split * a.taps * a.cout * a.cin * sizeof(float) : 0;
    SGV3D_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "conv2d_backward_weight_batched: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    a.x = x; a.dy = nullptr; a.dw = nullptr; a.ws = static_cast<float *>(workspace);
    a.nbatch = n;
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(dy_list[i] && dw_list[i] && ((uintptr_t)dy_list[i] & 15) == 0, "conv2d_backward_weight_batched: null / unaligned pointer %d", i);
        a.dy_list[i] = dy_list[i];
        a.dw_list[i] = dw_list[i];
    }
    hipStream_t st = as_stream(stream);
    conv_wgrad3x3_kernel<<<dim3(a.tiles_co * a.tiles_ci, a.split, n), 256, 0, st>>>(a);
    if (int rc = check_launch("conv_wgrad3x3_kernel")) return rc;
    if (a.split > 1) {
        const long long total = (long long)a.taps * a.cout * a.cin;
        wgrad_reduce_kernel<<<dim3(cdiv(total, 256), n), 256, 0, st>>>(a);
        return check_launch("wgrad_reduce_kernel");
    }
    return SGV3D_OK;
}

namespace {
int thin_fill(const sgv3d_conv_desc *d, ThinArgs &a) {
    SGV3D_REQUIRE(d, "conv2d_backward_weight_thin: null descriptor");
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil == 1 && d->pad >= 0 && d->cout >= 1 && d->cout <= 4 && d->y_ld <= 4,
                  "conv2d_backward_weight_thin: 3x3 / stride 1 / dilation 1 layers with 1..4 output channels and y_ld <= 4");
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout,
                  "conv2d_backward_weight_thin: bad sizes");
    SGV3D_REQUIRE(d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2, "conv2d_backward_weight_thin: output size does not belong to this input size");
    const unsigned long long xb = (unsigned long long)d->batch * d->in_h * d->in_w * d->x_ld * 4ull;
    const unsigned long long yb = (unsigned long long)d->batch * d->out_h * d->out_w * d->y_ld * 4ull;
    SGV3D_REQUIRE(xb < 0xf0000000ull && yb < 0xf0000000ull, "conv2d_backward_weight_thin: x / dy must be smaller than 3.75 GiB");
    a = ThinArgs{};
    a.batch = d->batch; a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w; a.pad = d->pad;
    a.cin = d->cin; a.cout = d->cout; a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.segs = cdiv(d->out_w, kThinSeg);
    a.units = d->batch * d->out_h * a.segs;
    int wgs = cdiv(a.units, 4);
    wgs = wgs > 512 ? 512 : wgs;          // two waves per SIMD; one partial set per workgroup
    a.waves = wgs * 4;
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
    return SGV3D_OK;
}
}  // namespace

extern "C" size_t sgv3d_conv2d_backward_weight_thin_workspace_bytes(const sgv3d_conv_desc *d) {
    ThinArgs a;
    if (thin_fill(d, a) != SGV3D_OK) return 0;
    return (size_t)(a.waves / 4) * 9 * a.cout * a.cin * sizeof(float);
}

extern "C" int sgv3d_conv2d_backward_weight_thin(const sgv3d_conv_desc *d, const float *x, const float *dy, float *dw, void *workspace,
                                                 size_t workspace_bytes, void *stream) {
    ThinArgs a;
    if (int rc = thin_fill(d, a)) return rc;
    SGV3D_REQUIRE(x && dy && dw && workspace, "conv2d_backward_weight_thin: null pointer");
    const size_t need = (size_t)(a.waves / 4) * 9 * a.cout * a.cin * sizeof(float);
    SGV3D_REQUIRE(workspace_bytes >= need, "conv2d_backward_weight_thin: workspace too small (%zu < %zu)", workspace_bytes, need);
    a.x = x; a.dy = dy; a.dw = dw; a.ws = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    const dim3 grid(a.waves / 4, cdiv(a.cin, 64));
    switch (a.cout) {
        case 1: wgrad_thin_kernel<1><<<grid, 256, 0, st>>>(a); break;
        case 2: wgrad_thin_kernel<2><<<grid, 256, 0, st>>>(a); break;
        case 3: wgrad_thin_kernel<3><<<grid, 256, 0, st>>>(a); break;
        default: wgrad_thin_kernel<4><<<grid, 256, 0, st>>>(a); break;
    }
    if (int rc = check_launch("wgrad_thin_kernel")) return rc;
    wgrad_thin_reduce_kernel<<<cdiv(9 * a.cout * a.cin, 256), 256, 0, st>>>(a);
    return check_launch("wgrad_thin_reduce_kernel");
}

extern "C" int sgv3d_zero_insert(int batch, int in_h, int in_w, int channels, int stride, int out_h, int out_w,
                                 const float *x, float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && channels > 0 && stride > 0, "zero_insert: bad sizes");
    SGV3D_REQUIRE(channels % 4 == 0, "zero_insert: channels must be a multiple of 4");
    SGV3D_REQUIRE(out_h >= (in_h - 1) * stride + 1 && out_w >= (in_w - 1) * stride + 1, "zero_insert: output too small");
    SGV3D_REQUIRE(x && y, "zero_insert: null pointer");
    const long long total = (long long)batch * out_h * out_w * (channels / 4);
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    zero_insert_kernel<<<blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y),
                                                             batch, in_h, in_w, channels / 4, stride, out_h, out_w);
    return check_launch("zero_insert_kernel");
}

extern "C" int sgv3d_interleave_phases2(int batch, int out_h, int out_w, int channels, const float *const *phases,
                                        const int32_t *phase_h, const int32_t *phase_w, const int32_t *row0,
                                        const int32_t *col0, float *y, void *stream) {
    SGV3D_REQUIRE(batch > 0 && out_h > 0 && out_w > 0 && channels > 0 && channels % 4 == 0, "interleave_phases2: bad sizes");
    SGV3D_REQUIRE(phases && phase_h && phase_w && row0 && col0 && y, "interleave_phases2: null pointer");
    PhaseArgs a{};
    for (int p = 0; p < 4; ++p) {
        const int need_h = (out_h - (p >> 1) + 1) / 2, need_w = (out_w - (p & 1) + 1) / 2;   // pixels of this phase
        SGV3D_REQUIRE(need_h == 0 || need_w == 0 || phases[p], "interleave_phases2: null phase");
        SGV3D_REQUIRE(row0[p] >= 0 && col0[p] >= 0 && row0[p] + need_h <= phase_h[p] && col0[p] + need_w <= phase_w[p],
                      "interleave_phases2: phase %d does not cover its pixels", p);
        a.src[p] = reinterpret_cast<const float4 *>(phases[p]);
        a.h[p] = phase_h[p]; a.w[p] = phase_w[p]; a.r0[p] = row0[p]; a.c0[p] = col0[p];
    }
    a.dst = reinterpret_cast<float4 *>(y);
    a.batch = batch; a.out_h = out_h; a.out_w = out_w; a.c4 = channels / 4;
    const long long total = (long long)batch * out_h * out_w * a.c4;
    interleave_phases_kernel<<<(int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192), 256, 0, as_stream(stream)>>>(a);
    return check_launch("interleave_phases_kernel");
}
