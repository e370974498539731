// Prints what gfx950's cross-lane instructions used by csrc/head_wino4.hip actually do (run on the GPU box):
//   v_permlane32_swap / v_permlane16_swap folds, DPP row_shr / row_shl with bound_ctrl, v_mfma_f32_16x16x4_f32 operand layout.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float *o) {
    const int l = threadIdx.x;
    const float a = (float)l, b = 100.f + l;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o[l] = __builtin_bit_cast(float, r[0]);
    o[64 + l] = __builtin_bit_cast(float, r[1]);
    const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o[128 + l] = __builtin_bit_cast(float, s[0]);
    o[192 + l] = __builtin_bit_cast(float, s[1]);
    o[256 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a + 1.f), 0x114, 0xf, 0xf, true));   // row_shr:4
    o[320 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a + 1.f), 0x104, 0xf, 0xf, true));   // row_shl:4
    o[384 + l] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a + 1.f), 0x111, 0xf, 0xf, true));   // row_shr:1
    // MFMA: A[i][k] = 1 if (i == probe row) ..., simpler: A[i][k] = i + 100 k (lane supplies one element), B[k][j] = delta(k, K0) * (j + 1)
    for (int K0 = 0; K0 < 4; ++K0) {
        const float av = (float)(l % 16) + 100.f * (l / 16);          // if A lane layout is i = l % 16, k = l / 16
        const float bv = (l / 16 == K0) ? (float)(l % 16 + 1) : 0.f;   // if B lane layout is k = l / 16, j = l % 16
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
        for (int r4 = 0; r4 < 4; ++r4) o[448 + K0 * 256 + r4 * 64 + l] = c[r4];
    }
}

int main() {
    float *d, h[448 + 1024];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"permlane32_swap(a=l, b=100+l)[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]",
                           "dpp row_shr:4 of (l+1)", "dpp row_shl:4 of (l+1)", "dpp row_shr:1 of (l+1)"};
    for (int t = 0; t < 7; ++t) {
        printf("%s:\n", names[t]);
        for (int l = 0; l < 64; ++l) printf("%4.0f%s", h[t * 64 + l], l % 16 == 15 ? "\n" : " ");
    }
    // D[i][j] expected = A[i][K0] * B[K0][j] = (i + 100 K0) * (j + 1)
    for (int K0 = 0; K0 < 4; ++K0) {
        printf("mfma K0=%d: per lane l (col j = l %% 16 if standard), regs r: value / (j + 1) should be i + 100 K0 with i = 4 (l / 16) + r\n", K0);
        for (int r4 = 0; r4 < 4; ++r4) {
            printf(" r=%d:", r4);
            for (int l = 0; l < 64; ++l) printf(" %5.0f", h[448 + K0 * 256 + r4 * 64 + l] / (l % 16 + 1));
            printf("\n");
        }
    }
    return 0;
}
