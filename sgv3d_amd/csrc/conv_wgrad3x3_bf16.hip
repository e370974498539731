// Weight gradient of a 3x3 / stride-1 convolution (any dilation) with ALL NINE TAPS in one workgroup, on the bf16 matrix cores
// (gfx950; mixed-precision training, BASELINE configs[4]; the reference leaves this to cuDNN's backward-filter behind
// nn.Conv2d, layers/backbones/lss_fpn.py:29-118, layers/heads/bev_height_head.py:75-110):
//
//   dW[co][ci][r][s] = sum over (image, oy, ox) of dY[image, oy, ox, co] * X[image, oy - pad + r dil, ox - pad + s dil, ci]
//
// The per-tap kernel (conv_wgrad_bf16_kernel) stages dY and X once per TAP: a 3x3 layer pulls both tensors nine times through
// L2 -> LDS, and that traffic, not the MFMA pipe, is what its 140-310 TFLOP/s are.  Here a workgroup owns a 64 (co) x 64 (ci)
// tile for all nine taps and walks DOWN a column of the output map (one 32-pixel segment wide): per output row it stages one
// new dY row segment and ONE new input row (the other two are still in LDS from the rows above: a ring of four row slots), so
// every input element is read from HBM / L2 once per column instead of nine times, and 18 MFMAs per wave follow each barrier.
//
// Operand layout: the reduction index of this GEMM is the PIXEL, which is the slow axis of both NHWC operands, and
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane.  The LDS images are kept in NHWC order -- [pixel][64 channels] bf16,
// 128 B per pixel, filled with plain 16-byte stores of 8 rounded channels -- and read with ds_read_b64_tr_b16, gfx950's
// transposing LDS read (a 4-pixel x 16-channel block per 16 lanes, delivered channel-major).  A tap's shift is then a ROW
// offset of the image (s dil pixels; r selects the ring slot): no shifted copies, no unaligned reads.  Bank conflicts: the 64-byte
// half of a pixel row is swapped when bit 1 of the pixel index is set, so the four rows of a block fall on the four 16-bank
// quarters whatever the shift.
//
// Dilation d: output rows are walked with stride d (d interleaved "phases" per column), so the three input rows of a step are
// again the last two plus one new one.  Partial sums per (column, row chunk) go to the workspace [item][tap][co][ci] and
// wgrad3x3_bf16_reduce_kernel adds them in item order (deterministic) into OIHW.
// Bound: MFMA bf16 (2.5 PFLOP/s dense); algorithmic 2 * 9 * pixels * cout * cin flop, HBM bytes = one read of X and dY per 64-wide
// tile of the other operand's channels.
#include "common.hpp"

using namespace sgv3d;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxBatch = 48;
constexpr int kP = 32;                   // pixels per output row segment = k per stage (two k-steps of 16)
constexpr int kMaxDil = 20;
constexpr int kXRows = kP + 2 * kMaxDil; // pixel rows of one ring slot
constexpr int kRowB = 128;               // bytes per pixel row of an LDS image (64 channels bf16)
constexpr int kDyBytes = kP * kRowB;     // 4096
constexpr int kXBytes = kXRows * kRowB;  // 9216
constexpr int kXPasses = (kXRows * 8 + 255) / 256;   // staging passes of 256 threads over (pixel, 8-channel chunk)

struct W3Args {
    const float *x, *dy;
    float *dw, *ws;
    const float *dy_list[kMaxBatch];
    float *dw_list[kMaxBatch];
    int nbatch;
    int batch, in_h, in_w, cin, out_h, out_w, cout, pad, dil;
    int x_ld, x_coff, y_ld, y_coff;
    int tiles_co, tiles_ci, segs, chunks, rows_per_chunk, items;
    unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ unsigned img_off(int row, int bytecol) { return (unsigned)(row * kRowB + (bytecol ^ (((row >> 1) & 1) << 6))); }

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char *lds, unsigned off) {
    // rows (pixels) k .. k + 3 and k + 4 .. k + 7 of this lane's channel: two transposed reads of 4 x 16 blocks
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lds + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lds + off + 4 * kRowB));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_bf16_kernel(const W3Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kDyBytes + 4 * kXBytes];
    unsigned char *const dyb = lds, *const xsb = lds + 2 * kDyBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    // item -> (image, segment, phase, chunk)
    int it = blockIdx.y;
    const int chunk = it % a.chunks; it /= a.chunks;
    const int phase = it % a.dil; it /= a.dil;
    const int seg = it % a.segs;
    const int img = it / a.segs;
    const int nj = phase < a.out_h ? (a.out_h - phase + a.dil - 1) / a.dil : 0;     // output rows of this phase
    const int j0 = chunk * a.rows_per_chunk;
    const int j1 = min(j0 + a.rows_per_chunk, nj);
    const int ox0 = seg * kP;
    const float *const dyp = a.nbatch ? a.dy_list[blockIdx.z] : a.dy;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dyp, 0, (int)a.y_bytes, 0x00020000);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- staging: a thread owns (pixel, 8-channel chunk) = tid / 8, tid % 8 (+ 32 pixels per further pass of the input row)
    const int sp = tid >> 3, sc = (tid & 7) * 8;
    const int sw = kP + 2 * a.dil;                        // staged pixels of an input row
    const int xpasses = (sw * 8 + 255) >> 8;
    // dY: pixel ox0 + sp, channels co0 + sc .. + 8 (two 16-byte halves, each in range or not: channel counts are multiples of 4)
    const bool dy_px_ok = ox0 + sp < a.out_w;
    const unsigned dy_c = (unsigned)(a.y_coff + co0 + sc) * 4u;
    const bool dy_ok0 = dy_px_ok && co0 + sc < a.cout, dy_ok1 = dy_px_ok && co0 + sc + 4 < a.cout;
    const unsigned x_c = (unsigned)(a.x_coff + ci0 + sc) * 4u;
    const bool x_ok0 = ci0 + sc < a.cin, x_ok1 = ci0 + sc + 4 < a.cin;

    f32x4 rdy[2], rx[kXPasses][2];
    auto load_dy = [&](int j) {                           // output row phase + j dil
        const int oy = phase + j * a.dil;
        const unsigned base = (unsigned)((img * a.out_h + oy) * a.out_w + ox0 + sp) * (unsigned)(a.y_ld * 4) + dy_c;
        const bool row_ok = j < j1;
        rdy[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, (row_ok && dy_ok0) ? base : 0xffffffffu, 0, 0));
        rdy[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, (row_ok && dy_ok1) ? base + 16u : 0xffffffffu, 0, 0));
    };
    auto load_x = [&](int v) {                            // virtual input row v: image row phase - pad + v dil
        const int iy = phase - a.pad + v * a.dil;
        const bool row_ok = (unsigned)iy < (unsigned)a.in_h;
#pragma unroll
        for (int p = 0; p < kXPasses; ++p) {
            if (p < xpasses) {
                const int t = sp + 32 * p;
                const int ix = ox0 - a.pad + t;
                const bool ok = row_ok && t < sw && (unsigned)ix < (unsigned)a.in_w;
                const unsigned base = (unsigned)((img * a.in_h + iy) * a.in_w + ix) * (unsigned)(a.x_ld * 4) + x_c;
                rx[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (ok && x_ok0) ? base : 0xffffffffu, 0, 0));
                rx[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (ok && x_ok1) ? base + 16u : 0xffffffffu, 0, 0));
            }
        }
    };
    auto store_dy = [&](int j) {
        const f32x8 v = __builtin_shufflevector(rdy[0], rdy[1], 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8 *>(dyb + (j & 1) * kDyBytes + img_off(sp, sc * 2)) = __builtin_convertvector(v, bf16x8);
    };
    auto store_x = [&](int v) {
        unsigned char *slot = xsb + (v & 3) * kXBytes;
#pragma unroll
        for (int p = 0; p < kXPasses; ++p) {
            if (p < xpasses) {
                const int t = sp + 32 * p;
                if (t < sw) {
                    const f32x8 w = __builtin_shufflevector(rx[p][0], rx[p][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    *reinterpret_cast<bf16x8 *>(slot + img_off(t, sc * 2)) = __builtin_convertvector(w, bf16x8);
                }
            }
        }
    };

    // ---- fragment addresses (ds_read_b64_tr_b16: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3 of the block)
    const int fq = (lane & 15) >> 2, fp = lane & 3;
    const int frow = fq + 8 * (lane >> 5);                               // k = 8 (lane / 32) + q (+ 4 for the second read)
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * fp;                    // channel within the wave's 32
    const unsigned a_off = img_off(frow, (wm * 32 + fcol) * 2);
    unsigned b_off[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) b_off[s] = img_off(frow + s * a.dil, (wn * 32 + fcol) * 2);

    if (j0 < j1) {
        // prologue: the first dY row and three input rows
        load_dy(j0);
        load_x(j0);
        store_dy(j0);
        store_x(j0);
        load_x(j0 + 1);
        store_x(j0 + 1);
        load_x(j0 + 2);
        store_x(j0 + 2);
        __syncthreads();
        for (int j = j0; j < j1; ++j) {
            // requests of the next step first (the row after the last one loads zeros for dY; its input row is never used)
            load_dy(j + 1);
            load_x(j + 3);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *const dyi = dyb + (j & 1) * kDyBytes;
#pragma unroll
            for (int ks = 0; ks < kP / 16; ++ks) {
                const bf16x8 fa = tr_read8(dyi, a_off + ks * 16 * kRowB);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const unsigned char *const xi = xsb + ((j + r) & 3) * kXBytes + ks * 16 * kRowB;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const bf16x8 fb = tr_read8(xi, b_off[s]);
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[r * 3 + s], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            store_dy(j + 1);
            store_x(j + 3);
            __syncthreads();
        }
    }
    // partial tile -> workspace [layer][item][tap][co][ci]
    const size_t plane = (size_t)a.cout * a.cin;
    float *ws = a.ws + ((size_t)blockIdx.z * a.items + blockIdx.y) * 9 * plane;
    const int ci = ci0 + wn * 32 + (lane & 31);
    if (ci < a.cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + wm * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
                if (co < a.cout) ws[(size_t)t * plane + (size_t)co * a.cin + ci] = acc[t][i];
            }
    }
}

// dw[co][ci][tap] = sum over the items (in order) of ws[layer][item][tap][co][ci]
__global__ __launch_bounds__(256) void wgrad3x3_bf16_reduce_kernel(const W3Args a) {
    const long long total = 9ll * a.cout * a.cin;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float *ws = a.ws + (size_t)blockIdx.y * a.items * total + i;
    float v = 0.f;
    int p = 0;
    for (; p + 8 <= a.items; p += 8) {                      // eight independent loads in flight, added in order
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ws[(size_t)(p + k) * total];
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
    }
    for (; p < a.items; ++p) v += ws[(size_t)p * total];
    const int ci = (int)(i % a.cin);
    const long long r = i / a.cin;
    const int co = (int)(r % a.cout), tap = (int)(r / a.cout);
    (a.nbatch ? a.dw_list[blockIdx.y] : a.dw)[((size_t)co * a.cin + ci) * 9 + tap] = v;
}

int fill(const sgv3d_conv_desc *d, int n, int split, W3Args &a) {
    SGV3D_REQUIRE(d, "conv2d_backward_weight_bf16_alltaps: null descriptor");
    SGV3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dil >= 1 && d->dil <= kMaxDil && d->pad >= 0,
                  "conv2d_backward_weight_bf16_alltaps: 3x3 / stride 1 layers with dilation 1 .. %d", kMaxDil);
    SGV3D_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0 && d->out_h > 0 && d->out_w > 0,
                  "conv2d_backward_weight_bf16_alltaps: bad sizes");
    SGV3D_REQUIRE(d->out_h == d->in_h + 2 * d->pad - 2 * d->dil && d->out_w == d->in_w + 2 * d->pad - 2 * d->dil,
                  "conv2d_backward_weight_bf16_alltaps: output size does not belong to this input size");
    SGV3D_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_coff % 4 == 0 && d->y_coff % 4 == 0 && d->x_ld % 4 == 0 && d->y_ld % 4 == 0 &&
                  d->x_ld >= d->x_coff + d->cin && d->y_ld >= d->y_coff + d->cout,
                  "conv2d_backward_weight_bf16_alltaps: channel counts, strides and offsets must be multiples of 4");
    const unsigned long long xb = (unsigned long long)d->batch * d->in_h * d->in_w * d->x_ld * 4ull;
    const unsigned long long yb = (unsigned long long)d->batch * d->out_h * d->out_w * d->y_ld * 4ull;
    SGV3D_REQUIRE(xb < 0xf0000000ull && yb < 0xf0000000ull, "conv2d_backward_weight_bf16_alltaps: x / dy must be smaller than 3.75 GiB");
    a = W3Args{};
    a.batch = d->batch; a.in_h = d->in_h; a.in_w = d->in_w; a.cin = d->cin; a.out_h = d->out_h; a.out_w = d->out_w; a.cout = d->cout;
    a.pad = d->pad; a.dil = d->dil; a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.tiles_co = cdiv(d->cout, 64); a.tiles_ci = cdiv(d->cin, 64);
    a.segs = cdiv(d->out_w, kP);
    const int cols = d->batch * a.segs * d->dil;
    const int nj = cdiv(d->out_h, d->dil);                 // rows of the longest phase
    int chunks = split;
    if (chunks <= 0) {
        // about three workgroups per CU over the launch (two are resident at a time), at least 8 rows per chunk (2 rows of prologue each)
        const long long wgs = (long long)n * a.tiles_co * a.tiles_ci * cols;
        chunks = (int)((768 + wgs - 1) / wgs);
    }
    const int max_chunks = nj / 8 > 0 ? nj / 8 : 1;
    chunks = chunks < 1 ? 1 : (chunks > max_chunks ? max_chunks : chunks);
    a.rows_per_chunk = cdiv(nj, chunks);
    a.chunks = cdiv(nj, a.rows_per_chunk);
    a.items = cols * a.chunks;
    SGV3D_REQUIRE(a.items <= 65535, "conv2d_backward_weight_bf16_alltaps: too many work items (%d)", a.items);
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
    return SGV3D_OK;
}

size_t ws_bytes(const W3Args &a, int n) { return (size_t)n * a.items * 9 * a.cout * a.cin * sizeof(float); }

int launch(W3Args &a, int n, hipStream_t st) {
    conv_wgrad3x3_bf16_kernel<<<dim3(a.tiles_co * a.tiles_ci, a.items, n), 256, 0, st>>>(a);
    if (int rc = check_launch("conv_wgrad3x3_bf16_kernel")) return rc;
    wgrad3x3_bf16_reduce_kernel<<<dim3((unsigned)cdiv(9ll * a.cout * a.cin, 256), n), 256, 0, st>>>(a);
    return check_launch("wgrad3x3_bf16_reduce_kernel");
}

}  // namespace

extern "C" size_t sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes(const sgv3d_conv_desc *d, int n, int split) {
    W3Args a;
    if (n < 1 || n > kMaxBatch || fill(d, n, split, a) != SGV3D_OK) return 0;
    return ws_bytes(a, n);
}

extern "C" int sgv3d_conv2d_backward_weight_bf16_alltaps(const sgv3d_conv_desc *d, const float *x, const float *dy, float *dw, int split,
                                                         void *workspace, size_t workspace_bytes, void *stream) {
    W3Args a;
    if (int rc = fill(d, 1, split, a)) return rc;
    SGV3D_REQUIRE(x && dy && dw && workspace, "conv2d_backward_weight_bf16_alltaps: null pointer");
    SGV3D_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv2d_backward_weight_bf16_alltaps: x / dy must be 16-byte aligned");
    SGV3D_REQUIRE(workspace_bytes >= ws_bytes(a, 1), "conv2d_backward_weight_bf16_alltaps: workspace too small (%zu < %zu)", workspace_bytes, ws_bytes(a, 1));
    a.x = x; a.dy = dy; a.dw = dw; a.ws = static_cast<float *>(workspace);
    return launch(a, 1, as_stream(stream));
}

extern "C" int sgv3d_conv2d_backward_weight_bf16_alltaps_batched(const sgv3d_conv_desc *d, const float *x, const float *const *dy_list,
                                                                 float *const *dw_list, int n, int split, void *workspace,
                                                                 size_t workspace_bytes, void *stream) {
    SGV3D_REQUIRE(d && x && dy_list && dw_list && n > 0 && n <= kMaxBatch, "conv2d_backward_weight_bf16_alltaps_batched: 1 .. %d problems", kMaxBatch);
    W3Args a;
    if (int rc = fill(d, n, split, a)) return rc;
    SGV3D_REQUIRE(workspace && ((uintptr_t)x & 15) == 0, "conv2d_backward_weight_bf16_alltaps_batched: null workspace / unaligned x");
    SGV3D_REQUIRE(workspace_bytes >= ws_bytes(a, n), "conv2d_backward_weight_bf16_alltaps_batched: workspace too small (%zu < %zu)", workspace_bytes, ws_bytes(a, n));
    a.x = x; a.ws = static_cast<float *>(workspace);
    a.nbatch = n;
    for (int i = 0; i < n; ++i) {
        SGV3D_REQUIRE(dy_list[i] && dw_list[i] && ((uintptr_t)dy_list[i] & 15) == 0, "conv2d_backward_weight_bf16_alltaps_batched: null / unaligned pointer %d", i);
        a.dy_list[i] = dy_list[i];
        a.dw_list[i] = dw_list[i];
    }
    return launch(a, n, as_stream(stream));
}
