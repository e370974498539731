// Micro-benchmark: do VALU instructions of ONE wave overlap with f32 MFMAs of ANOTHER wave on the same SIMD?
// A workgroup is 8 waves = 2 per SIMD.  Waves 0-3 run a pure v_mfma_f32_32x32x2_f32 loop; waves 4-7 run nothing (mode 0), a
// pure v_fma_f32 loop of about the same issue time (mode 1), or a v_pk_fma_f32 loop (mode 2).  If the two kinds execute on
// different pipes the launch takes as long as the longer of the two loops; if they share the SIMD's FP32 lanes it takes the sum.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, float av, float bv, int mfma_on) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!mfma_on) return;
        f32x16 acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
        const float a = av + threadIdx.x, b = bv;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) { acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0); SB(); }
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) s += acc[p][0] + acc[p][15];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (MODE == 0) return;
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = av + i + threadIdx.x;
        const float m = bv;
        if (MODE == 1) {
            // 16 MFMAs of the other wave = 1024 cycles; a wave64 v_fma_f32 issues in 4 cycles -> 256 of them
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], m, 1.0f);
                    SB();
                }
            }
        } else {
            f32x2 y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = f32x2{x[2 * i], x[2 * i + 1]};
            const f32x2 mm = {m, m}, one = {1.f, 1.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], mm, one);
                    SB();
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[2 * i] = y[i][0]; x[2 * i + 1] = y[i][1]; }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int MODE>
float run(float *out, int iters, int mfma_on) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(out, iters, 1.0f, 0.999f, mfma_on);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(out, iters, 1.0f, 0.999f, mfma_on);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    printf("MFMA waves alone                : %8.1f us\n", run<0>(out, iters, 1));
    printf("v_fma_f32 waves alone           : %8.1f us\n", run<1>(out, iters, 0));
    printf("v_pk_fma_f32 waves alone        : %8.1f us\n", run<2>(out, iters, 0));
    printf("MFMA + v_fma_f32 on the same SIMD   : %8.1f us\n", run<1>(out, iters, 1));
    printf("MFMA + v_pk_fma_f32 on the same SIMD: %8.1f us\n", run<2>(out, iters, 1));
    return 0;
}
