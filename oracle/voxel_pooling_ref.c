/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (sgv3d_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Plain-C restatement of the reference's voxel-pooling operator for CPU:
 *
 *   forward  : voxel_pooling_forward_kernel
 *              /root/reference/ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36
 *              (bounds test :24, pos_memo :27-29, channel loop of atomicAdd :30-33)
 *   backward : VoxelPooling.backward
 *              /root/reference/ops/voxel_pooling/voxel_pooling.py:58-69
 *              (grad_in[p,:] = grad_out[b,:,y,x] for kept points, 0 elsewhere)
 *
 * The reference has no CPU implementation of this op (its extension is CUDA-only,
 * voxel_pooling_forward.cpp:2-4) and cannot be compiled in this image (THC/THC.h, cuda.h), so
 * there is no oracle/_ref build: this restatement is pinned against golden vectors captured by
 * running the reference's own Python wrapper (VoxelPooling.apply, voxel_pooling.py:10-55) around
 * a kernel stub that follows the 28-line .cu literally (tests/golden/make_golden.py).
 *
 * Summation order: the CUDA kernel's float atomics are order-nondeterministic; this restatement
 * adds in ascending point order (p = 0..B*N-1), single thread => deterministic.  On integer-valued
 * floats every order gives the same bits, which is what the exact parity tests use.
 *
 * The _omp variant splits the *voxel rows* (y) across threads so each output element still sees its
 * points in ascending order: bit-identical to the single-thread version, used only for the
 * cpu_baseline timing with cores > 1.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* geom_xyz int32 [B*N,3]; feats f32 [B*N,C]; out f32 [B,Y,X,C] (accumulated in place, caller
 * pre-zeroes like voxel_pooling.py:37-38); pos_memo int32 [B*N,3] (caller pre-fills -1, :40). */
void sgv3d_oracle_voxel_pooling_forward(int batch_size, int num_points, int num_channels,
                                        int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                        const int32_t *geom_xyz, const float *input_features,
                                        float *output_features, int32_t *pos_memo)
{
    const long total = (long)batch_size * num_points;
    for (long pt = 0; pt < total; ++pt) {
        const int batch_idx = (int)(pt / num_points);
        const int x = geom_xyz[pt * 3 + 0];
        const int y = geom_xyz[pt * 3 + 1];
        const int z = geom_xyz[pt * 3 + 2];
        if (x < 0 || x >= num_voxel_x || y < 0 || y >= num_voxel_y || z < 0 || z >= num_voxel_z)
            continue;
        if (pos_memo) {
            pos_memo[pt * 3 + 0] = batch_idx;
            pos_memo[pt * 3 + 1] = y;
            pos_memo[pt * 3 + 2] = x;
        }
        float *dst = output_features +
                     ((size_t)batch_idx * num_voxel_y * num_voxel_x + (size_t)y * num_voxel_x + x) *
                         num_channels;
        const float *src = input_features + (size_t)pt * num_channels;
        for (int c = 0; c < num_channels; ++c)
            dst[c] += src[c];
    }
}

/* Same result bit for bit on all host cores (cpu_baseline, cores > 1).  The points are partitioned ONCE: every thread scans
 * its own contiguous chunk of geom_xyz, the kept points are bucketed by the thread that owns their voxel row (y mod threads;
 * a stable counting sort, so every voxel still sees its points in ascending point order = the scalar loop's summation
 * order), and each thread then sums only its own bucket.  (The round-1 form had every thread scan all points.) */
void sgv3d_oracle_voxel_pooling_forward_omp(int batch_size, int num_points, int num_channels,
                                            int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                            const int32_t *geom_xyz, const float *input_features,
                                            float *output_features, int32_t *pos_memo,
                                            int num_threads)
{
#ifdef _OPENMP
    if (num_threads < 1) num_threads = 1;
    const long total = (long)batch_size * num_points;
    const int nth = num_threads;
    long *cnt = (long *)calloc((size_t)nth * nth + 1, sizeof(long));      /* cnt[scanner][owner] -> offsets */
    int32_t *list = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    long *begin = (long *)calloc((size_t)nth + 1, sizeof(long));
    if (!cnt || !list || !begin || total > 0x7fffffffL) {                 /* (out of memory / too many points: scalar loop) */
        free(cnt); free(list); free(begin);
        sgv3d_oracle_voxel_pooling_forward(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y, num_voxel_z,
                                           geom_xyz, input_features, output_features, pos_memo);
        return;
    }
#pragma omp parallel num_threads(nth)
    {
        const int tid = omp_get_thread_num();
        const int n = omp_get_num_threads();       /* (may be fewer than asked for: chunks and owners use nth, looped over) */
        for (int s = tid; s < nth; s += n) {       /* pass 1: count the kept points of chunk s by owner */
            const long p0 = total * s / nth, p1 = total * (s + 1) / nth;
            long *c = cnt + (size_t)s * nth;
            for (long pt = p0; pt < p1; ++pt) {
                const int x = geom_xyz[pt * 3 + 0], y = geom_xyz[pt * 3 + 1], z = geom_xyz[pt * 3 + 2];
                if (x < 0 || x >= num_voxel_x || y < 0 || y >= num_voxel_y || z < 0 || z >= num_voxel_z) continue;
                c[y % nth]++;
            }
        }
#pragma omp barrier
#pragma omp single
        {                                          /* offsets: owner-major, scanner-minor = ascending point order per owner */
            long run = 0;
            for (int o = 0; o < nth; ++o) {
                begin[o] = run;
                for (int s = 0; s < nth; ++s) {
                    const long k = cnt[(size_t)s * nth + o];
                    cnt[(size_t)s * nth + o] = run;
                    run += k;
                }
            }
            begin[nth] = run;
        }
        for (int s = tid; s < nth; s += n) {       /* pass 2: fill the buckets, write pos_memo */
            const long p0 = total * s / nth, p1 = total * (s + 1) / nth;
            long *c = cnt + (size_t)s * nth;
            for (long pt = p0; pt < p1; ++pt) {
                const int x = geom_xyz[pt * 3 + 0], y = geom_xyz[pt * 3 + 1], z = geom_xyz[pt * 3 + 2];
                if (x < 0 || x >= num_voxel_x || y < 0 || y >= num_voxel_y || z < 0 || z >= num_voxel_z) continue;
                list[c[y % nth]++] = (int32_t)pt;
                if (pos_memo) {
                    pos_memo[pt * 3 + 0] = (int)(pt / num_points);
                    pos_memo[pt * 3 + 1] = y;
                    pos_memo[pt * 3 + 2] = x;
                }
            }
        }
#pragma omp barrier
        for (int o = tid; o < nth; o += n) {       /* pass 3: every owner sums its own voxel rows */
            for (long i = begin[o]; i < begin[o + 1]; ++i) {
                const long pt = list[i];
                const int x = geom_xyz[pt * 3 + 0], y = geom_xyz[pt * 3 + 1];
                const int batch_idx = (int)(pt / num_points);
                float *dst = output_features + ((size_t)batch_idx * num_voxel_y * num_voxel_x +
                                                (size_t)y * num_voxel_x + x) * num_channels;
                const float *src = input_features + (size_t)pt * num_channels;
                for (int c = 0; c < num_channels; ++c)
                    dst[c] += src[c];
            }
        }
    }
    free(cnt); free(list); free(begin);
#else
    (void)num_threads;
    sgv3d_oracle_voxel_pooling_forward(batch_size, num_points, num_channels, num_voxel_x,
                                       num_voxel_y, num_voxel_z, geom_xyz, input_features,
                                       output_features, pos_memo);
#endif
}

/* grad_out f32 [B,C,Y,X] (contiguous NCHW, what autograd hands VoxelPooling.backward);
 * grad_in f32 [B*N,C] fully written (zeros for dropped points, voxel_pooling.py:29,64). */
void sgv3d_oracle_voxel_pooling_backward(int batch_size, int num_points, int num_channels,
                                         int num_voxel_x, int num_voxel_y,
                                         const int32_t *pos_memo, const float *grad_output,
                                         float *grad_input)
{
    const long total = (long)batch_size * num_points;
    const size_t plane = (size_t)num_voxel_y * num_voxel_x;
    for (long pt = 0; pt < total; ++pt) {
        float *dst = grad_input + (size_t)pt * num_channels;
        const int b = pos_memo[pt * 3 + 0];
        if (b == -1) { /* kept = (pos_memo != -1)[..., 0], voxel_pooling.py:60 */
            memset(dst, 0, sizeof(float) * (size_t)num_channels);
            continue;
        }
        const int y = pos_memo[pt * 3 + 1];
        const int x = pos_memo[pt * 3 + 2];
        const float *src = grad_output + (size_t)b * num_channels * plane + (size_t)y * num_voxel_x + x;
        for (int c = 0; c < num_channels; ++c)
            dst[c] = src[(size_t)c * plane];
    }
}
