// Fused AdamW update over a flat fp32 parameter bucket (the optimiser step of
// exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:298-305, torch.optim.AdamW(lr, weight_decay=1e-7)).
// One pass over HBM: reads p, g, m, v (16 B per parameter), writes p, m, v (12 B); the gradient average over the
// data-parallel ranks (1 / world size) is folded in as grad_scale, so the all-reduced sum needs no pass of its own.
// Arithmetic follows torch's single-tensor AdamW step operation by operation (decoupled decay first, bias
// corrections as step_size = lr / (1 - beta1^t) and denom = sqrt(v) / sqrt(1 - beta2^t) + eps).
#include "common.hpp"

using namespace sgv3d;

namespace {

struct AdamArgs {
    float *p, *m, *v;
    const float *g;
    long long n;
    float lr, beta1, beta2, eps, wd, grad_scale, step_size, inv_sqrt_bc2;
    const float *hyper;      // device [lr, step_size, inv_sqrt_bc2] read at kernel start instead of the three host scalars (sgv3d_adamw_step_dev)
    const float *clip;       // device scalar min(1, max_norm / (||g|| + 1e-6)) of sgv3d_clip_coef, or null: no gradient clipping
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a) {
    g *= a.grad_scale;
    p *= 1.f - a.lr * a.wd;
    m = m * a.beta1 + g * (1.f - a.beta1);          // lerp(m, g, 1 - beta1)
    v = v * a.beta2 + (g * g) * (1.f - a.beta2);
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    if (a.hyper) { a.lr = a.hyper[0]; a.step_size = a.hyper[1]; a.inv_sqrt_bc2 = a.hyper[2]; }
    if (a.clip) a.grad_scale *= a.clip[0];
    const long long n4 = a.n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(a.p), *m4 = reinterpret_cast<float4 *>(a.m), *v4 = reinterpret_cast<float4 *>(a.v);
    const float4 *g4 = reinterpret_cast<const float4 *>(a.g);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    for (long long i = n4 * 4 + blockIdx.x * 256ll + threadIdx.x; i < a.n; i += gridDim.x * 256ll)
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
}

constexpr int SUMSQ_PARTIALS = 1024;

// Sum of squares of a flat gradient bucket: block b owns the b-th of SUMSQ_PARTIALS equal contiguous slices and leaves ONE double
// (fixed slice, fixed in-block order: bitwise repeatable, whatever else runs on the chip).
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *g, long long n, double *partials) {
    const long long per = ((n + SUMSQ_PARTIALS - 1) / SUMSQ_PARTIALS + 3) / 4 * 4;
    const long long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    if (lo < hi) {
        const long long n4 = (hi - lo) / 4;
        const float4 *g4 = reinterpret_cast<const float4 *>(g + lo);
        for (long long i = threadIdx.x; i < n4; i += 256) {
            const float4 v = g4[i];
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (long long i = lo + n4 * 4 + threadIdx.x; i < hi; i += 256) acc += (double)g[i] * g[i];
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// total_norm = sqrt(sum of the partials) * grad_scale (the norm of the AVERAGED gradient: the buckets hold the all-reduced sum);
// out[0] = min(1, max_norm / (total_norm + 1e-6)) -- torch.nn.utils.clip_grad_norm_'s coefficient --, out[1] = total_norm.
__global__ __launch_bounds__(256) void clip_coef_kernel(const double *partials, int count, float grad_scale, float max_norm, float *out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)(sqrt(red[0]) * (double)grad_scale);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = coef < 1.f ? coef : 1.f;          // (a NaN norm gives a NaN coefficient, as torch's clamp(max=1) does)
        if (coef != coef) out[0] = coef;
        out[1] = norm;
    }
}

__global__ void set_hyper_kernel(float *hyper, float lr, float step_size, float inv_sqrt_bc2) {
    hyper[0] = lr; hyper[1] = step_size; hyper[2] = inv_sqrt_bc2;
}

}  // namespace

extern "C" int sgv3d_grad_sumsq_partials(void) { return SUMSQ_PARTIALS; }

// Global-norm gradient clipping of the training step (Lightning's ``gradient_clip_val=5``,
// exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:405 -> torch.nn.utils.clip_grad_norm_(parameters, 5)): one
// sgv3d_grad_sumsq per flat bucket into consecutive groups of sgv3d_grad_sumsq_partials() doubles, then ONE sgv3d_clip_coef over all
// of them; the AdamW entries read the coefficient from device memory (no host round trip: the sequence records into a hipGraph).
extern "C" int sgv3d_grad_sumsq(long long n, const float *grad, double *partials, void *stream) {
    SGV3D_REQUIRE(n >= 0 && partials, "grad_sumsq: bad n / null partials");
    SGV3D_REQUIRE(n == 0 || grad, "grad_sumsq: null gradient");
    SGV3D_REQUIRE(((uintptr_t)grad & 15) == 0, "grad_sumsq: the bucket must be 16-byte aligned");
    grad_sumsq_kernel<<<SUMSQ_PARTIALS, 256, 0, as_stream(stream)>>>(grad, n, partials);
    return check_launch("grad_sumsq_kernel");
}

extern "C" int sgv3d_clip_coef(const double *partials, int count, float grad_scale, float max_norm, float *out, void *stream) {
    SGV3D_REQUIRE(partials && out && count > 0, "clip_coef: null pointer / empty");
    SGV3D_REQUIRE(max_norm > 0.f, "clip_coef: max_norm must be positive");
    clip_coef_kernel<<<1, 256, 0, as_stream(stream)>>>(partials, count, grad_scale, max_norm, out);
    return check_launch("clip_coef_kernel");
}

// hyper[0..2] = [lr, lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)], computed on the host exactly as sgv3d_adamw_step computes them
// and handed over as KERNEL ARGUMENTS (copied at launch): a caller may stage step t + 1 while the update of step t is still queued.
extern "C" int sgv3d_adamw_set_hyper(float *hyper, int step, float lr, float beta1, float beta2, void *stream) {
    SGV3D_REQUIRE(hyper && step >= 1, "adamw_set_hyper: null pointer / bad step");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    set_hyper_kernel<<<1, 1, 0, as_stream(stream)>>>(hyper, lr, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
    return check_launch("set_hyper_kernel");
}

extern "C" int sgv3d_adamw_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float grad_scale, const float *clip_coef, void *stream) {
    SGV3D_REQUIRE(n >= 0 && step >= 1, "adamw_step: bad n / step");
    if (n == 0) return SGV3D_OK;
    SGV3D_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adamw_step: null pointer");
    SGV3D_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                  "adamw_step: buffers must be 16-byte aligned");
    AdamArgs a{};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.clip = clip_coef;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.grad_scale = grad_scale;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long long blocks = (n / 4 + 255) / 256 + 1;
    adamw_kernel<<<(int)(blocks < 4096 ? blocks : 4096), 256, 0, as_stream(stream)>>>(a);
    return check_launch("adamw_kernel");
}

// The same update with the step-dependent scalars in DEVICE memory: hyper = [lr, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)] (f32,
// what sgv3d_adamw_step computes on the host from step and lr).  A launch recorded in a hipGraph (train_step.GraphedTrainStep) then
// follows the step counter and the learning-rate schedule through one sgv3d_adamw_set_hyper launch per replay.  Bitwise sgv3d_adamw_step.
extern "C" int sgv3d_adamw_step_dev(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                    const float *hyper, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                    const float *clip_coef, void *stream) {
    SGV3D_REQUIRE(n >= 0, "adamw_step_dev: bad n");
    if (n == 0) return SGV3D_OK;
    SGV3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper, "adamw_step_dev: null pointer");
    SGV3D_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                  "adamw_step_dev: buffers must be 16-byte aligned");
    AdamArgs a{};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.hyper = hyper; a.clip = clip_coef;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.grad_scale = grad_scale;
    const long long blocks = (n / 4 + 255) / 256 + 1;
    adamw_kernel<<<(int)(blocks < 4096 ? blocks : 4096), 256, 0, as_stream(stream)>>>(a);
    return check_launch("adamw_kernel");
}
