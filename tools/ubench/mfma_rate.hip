// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate of ONE wave per SIMD with 16 accumulator tiles
// (256 accumulator registers), for different issue patterns.  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, float av, float bv, const float4 *gsrc) {
    __shared__ float4 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 ld = make_float4(0, 0, 0, 0);
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    float a = av + threadIdx.x, b = bv;
    float x0 = a, x1 = b, x2 = a * 2, x3 = b * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; p += 2) {
            if (MODE == 0) {          // 4 dependent back to back per accumulator
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB(); }
            } else if (MODE == 1) {   // two accumulators alternating
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB(); }
            } else if (MODE == 3) {   // alternating + 4 independent VALU (not feeding the MFMA) per MFMA
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB();
                        x0 = x0 - a; x1 = x1 + b; x2 = x2 - a; x3 = x3 - b; SB();
                    }
            } else if (MODE == 4) {   // alternating + 1 independent VALU per MFMA
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB();
                        x0 = x0 - a; SB();
                    }
            } else if (MODE == 5) {   // alternating + 8 independent VALU per MFMA
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB();
                        x0 = x0 - a; x1 = x1 + b; x2 = x2 - a; x3 = x3 - b; SB();
                        x0 = x0 * a; x1 = x1 * b; x2 = x2 * a; x3 = x3 * b; SB();
                    }
            } else if (MODE == 6) {   // VALU results feed the NEXT-but-one MFMA through ping-pong registers
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(q ? x2 : x0, q ? x3 : x1, acc[p + q], 0, 0, 0); SB();
                        if (q) { x0 = x0 - a; x1 = x1 + b; } else { x2 = x2 - a; x3 = x3 + b; }
                        SB();
                    }
            } else if (MODE == 7) {   // 8 MFMAs clean, then 32 independent VALU in one gap (= 4 per MFMA)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB(); }
#pragma unroll
                for (int r = 0; r < 8; ++r) { x0 = x0 - a; x1 = x1 + b; x2 = x2 - a; x3 = x3 - b; }
                SB();
            } else if (MODE == 8) {   // one ds_read_b128 per MFMA gap
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB();
                        ld = sm[(threadIdx.x + j * 64 + q * 32 + p * 8) & 1023]; SB();
                        x0 += ld.x;  SB();
                    }
            } else if (MODE == 9) {   // one global_load_dwordx4 per MFMA gap
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p + q], 0, 0, 0); SB();
                        ld = gsrc[(threadIdx.x + (j * 2 + q + p * 8) * 256) & 16383]; SB();
                        x0 += ld.x;  SB();
                    }
            } else {                  // alternating + 4 VALU per MFMA
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[p + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc[p + q], 0, 0, 0); SB();
                        x0 = x0 - x2; x1 = x1 + x3; x2 = x2 - x1; x3 = x3 - x0; SB();
                    }
            }
        }
    }
    float s = x0 + x1 + x2 + x3 + ld.y;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[p][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int blocks) {
    float *out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    float4 *g; hipMalloc(&g, 16384 * 16); hipMemset(g, 0, 16384 * 16);
    k<MODE><<<blocks, 256>>>(out, 10, 1.f, 2.f, g);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.f, 2.f, g);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 64;
    printf("%-40s blocks %4d: %8.3f ms  %6.1f ns/MFMA/wave  %7.1f TF\n", name, blocks, ms, ms * 1e6 / mfma_per_wave,
           mfma_per_wave * 4 * blocks * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int blocks : {256}) {
        run<0>("4 dependent back-to-back", blocks);
        run<1>("2 accumulators alternating", blocks);
        run<2>("alternating + 4 dependent VALU feeding it", blocks);
        run<3>("alternating + 4 independent VALU", blocks);
        run<4>("alternating + 1 independent VALU", blocks);
        run<5>("alternating + 8 independent VALU", blocks);
        run<6>("alternating + 2 VALU feeding MFMA +2", blocks);
        run<7>("8 clean MFMAs then 32 VALU in one gap", blocks);
        run<8>("one ds_read_b128 (+1 VALU) per gap", blocks);
        run<9>("one global_load_dwordx4 (+1 VALU) per gap", blocks);
    }
    return 0;
}
