// CenterHead target assignment on the device.
// Reference: BEVHeightHead.get_targets_single (layers/heads/bev_height_head.py:113-253) under mmdet3d 0.18.1
// CenterHead.get_targets, with gaussian_radius / draw_heatmap_gaussian of mmdet3d/core/utils/gaussian.py.
// The reference walks the boxes of a sample in Python, one device scalar at a time (a few dozen
// synchronising tensor ops per box); here one launch covers every (sample, box):
//
//   targets_kernel   one 64-lane workgroup per (box, sample).  The slot of a box inside its task is its
//                    position in the reference's regrouped list (all boxes of the task's first class in input
//                    order, then the second class, ...), computed as a rank count over the sample's labels.
//                    The lanes then evaluate the same fp32 expressions as the reference, in the same order
//                    (this file is compiled with -ffp-contract=off), lane 0 writes ind / mask / anno_box and
//                    all lanes splat the Gaussian window with an integer atomicMax on the float bits
//                    (values are >= 0, so the order of the maxima is irrelevant: deterministic).
//   zero kernel      outputs are cleared by a kernel, not hipMemsetAsync (memset nodes of captured graphs
//                    fault on replay on this stack).
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kMaxClasses = 32;
constexpr int kMaxTasks = 16;

struct TargetsArgs {
    const float *boxes;      // [B][n_max][9]
    const int32_t *labels;   // [B][n_max]
    float *heatmap;          // [B][total_classes][h][w]
    float *anno;             // [T][B][max_objs][10]
    long long *ind;          // [T][B][max_objs]
    unsigned char *mask;     // [T][B][max_objs]
    int batch, n_max, num_tasks, total_classes, max_objs, h, w, min_radius, norm_bbox;
    float pc_x, pc_y, voxel_x, voxel_y, osf;
    float k_1m, k_1p, k_m2, k_m1, k_16;   // f32(1-mo), f32(1+mo), f32(-2*mo), f32(mo-1), f32(4*(4*mo))
    signed char task_of[kMaxClasses];
};

struct ZeroArgs {
    unsigned char *p[4];
    unsigned long long n[4];
};

__global__ void __launch_bounds__(256) zero_segments_kernel(ZeroArgs a) {
    for (int s = 0; s < 4; ++s) {
        unsigned char *p = a.p[s];
        const unsigned long long n = a.n[s];
        const bool aligned = (reinterpret_cast<unsigned long long>(p) & 15) == 0;
        const unsigned long long n16 = aligned ? n / 16 : 0;
        uint4 *p16 = reinterpret_cast<uint4 *>(p);
        for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull)
            p16[i] = make_uint4(0, 0, 0, 0);
        for (unsigned long long i = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
            p[i] = 0;
    }
}

__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// mmdet3d gaussian_radius on fp32 scalars (det_size = (length, width)); r3 keeps upstream's "/ 2".
__device__ float gaussian_radius_f32(float height, float width, const TargetsArgs &a) {
    const float b1 = height + width;
    const float c1 = ((width * height) * a.k_1m) / a.k_1p;
    const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
    const float b2 = 2.f * (height + width);
    const float c2 = (a.k_1m * width) * height;
    const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
    const float b3 = a.k_m2 * (height + width);
    const float c3 = (a.k_m1 * width) * height;
    const float r3 = (b3 + sqrtf(b3 * b3 - a.k_16 * c3)) / 2.f;
    return fminf(r1, fminf(r2, r3));
}

__global__ void __launch_bounds__(64) targets_kernel(TargetsArgs a) {
    const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int32_t *lab = a.labels + (size_t)b * a.n_max;
    const int label = lab[i];
    if (label < 0 || label >= a.total_classes) return;
    const int t = a.task_of[label];
    // slot = number of boxes of the same task that the reference lists before this one
    int before = 0;
    for (int j = lane; j < a.n_max; j += 64) {
        const int lj = lab[j];
        if (lj >= 0 && lj < a.total_classes && a.task_of[lj] == t && (lj < label || (lj == label && j < i))) ++before;
    }
    const int k = wave_sum(before);
    if (k >= a.max_objs) return;
    const float *box = a.boxes + ((size_t)b * a.n_max + i) * 9;
    const float width = box[3] / a.voxel_x / a.osf;
    const float length = box[4] / a.voxel_y / a.osf;
    if (!(width > 0.f && length > 0.f)) return;
    const float rad = gaussian_radius_f32(length, width, a);
    int radius = (int)rad;                      // int(tensor): truncation
    radius = radius > a.min_radius ? radius : a.min_radius;
    const float coor_x = (box[0] - a.pc_x) / a.voxel_x / a.osf;
    const float coor_y = (box[1] - a.pc_y) / a.voxel_y / a.osf;
    const int cx = (int)coor_x, cy = (int)coor_y;   // .to(torch.int32): toward zero, so (-1, 0) lands in cell 0
    if (!(cx >= 0 && cx < a.w && cy >= 0 && cy < a.h)) return;
    if (lane == 0) {
        const size_t slot = ((size_t)t * a.batch + b) * a.max_objs + k;
        a.ind[slot] = (long long)cy * a.w + cx;
        a.mask[slot] = 1;
        float *ab = a.anno + slot * 10;
        ab[0] = coor_x - (float)cx;
        ab[1] = coor_y - (float)cy;
        ab[2] = box[2];
        ab[3] = a.norm_bbox ? logf(box[3]) : box[3];
        ab[4] = a.norm_bbox ? logf(box[4]) : box[4];
        ab[5] = a.norm_bbox ? logf(box[5]) : box[5];
        ab[6] = sinf(box[6]);
        ab[7] = cosf(box[6]);
        ab[8] = box[7];
        ab[9] = box[8];
    }
    // draw_heatmap_gaussian: float64 exp(-(dx^2 + dy^2) / (2 sigma^2)), sigma = diameter / 6, cast to fp32, max-merged
    int *hm = reinterpret_cast<int *>(a.heatmap + ((size_t)b * a.total_classes + label) * a.h * a.w);
    const int d = 2 * radius + 1;
    const double sigma = (double)d / 6.0;
    const double den = 2.0 * sigma * sigma;
    for (int e = lane; e < d * d; e += 64) {
        const int dy = e / d - radius, dx = e % d - radius;
        const int px = cx + dx, py = cy + dy;
        if (px < 0 || px >= a.w || py < 0 || py >= a.h) continue;
        const double xx = (double)dx, yy = (double)dy;
        const float g = (float)exp(-(xx * xx + yy * yy) / den);
        atomicMax(hm + (size_t)py * a.w + px, __float_as_int(g));
    }
}

}  // namespace

extern "C" int sgv3d_centerhead_targets(int batch, int n_max, const float *boxes, const int32_t *labels,
                                        int num_tasks, const int32_t *classes_per_task, int max_objs, int h,
                                        int w, float pc_x, float pc_y, float voxel_x, float voxel_y,
                                        float out_size_factor, double gaussian_overlap, int min_radius,
                                        int norm_bbox, float *heatmap, float *anno_box, long long *ind,
                                        unsigned char *mask, void *stream) {
    SGV3D_REQUIRE(batch > 0 && n_max >= 0 && max_objs > 0 && h > 0 && w > 0, "centerhead_targets: bad sizes");
    SGV3D_REQUIRE(num_tasks > 0 && num_tasks <= kMaxTasks && classes_per_task, "centerhead_targets: 1..%d tasks", kMaxTasks);
    SGV3D_REQUIRE(heatmap && anno_box && ind && mask, "centerhead_targets: null output");
    SGV3D_REQUIRE(n_max == 0 || (boxes && labels), "centerhead_targets: null input");
    SGV3D_REQUIRE(voxel_x > 0 && voxel_y > 0 && out_size_factor > 0, "centerhead_targets: bad cell size");
    SGV3D_REQUIRE(batch <= 65535, "centerhead_targets: batch > 65535");
    TargetsArgs a{};
    int total = 0;
    for (int t = 0; t < num_tasks; ++t) {
        SGV3D_REQUIRE(classes_per_task[t] > 0 && total + classes_per_task[t] <= kMaxClasses,
                      "centerhead_targets: at most %d classes", kMaxClasses);
        for (int c = 0; c < classes_per_task[t]; ++c) a.task_of[total + c] = (signed char)t;
        total += classes_per_task[t];
    }
    a.boxes = boxes; a.labels = labels; a.heatmap = heatmap; a.anno = anno_box; a.ind = ind; a.mask = mask;
    a.batch = batch; a.n_max = n_max; a.num_tasks = num_tasks; a.total_classes = total; a.max_objs = max_objs;
    a.h = h; a.w = w; a.min_radius = min_radius; a.norm_bbox = norm_bbox;
    a.pc_x = pc_x; a.pc_y = pc_y; a.voxel_x = voxel_x; a.voxel_y = voxel_y; a.osf = out_size_factor;
    const double mo = gaussian_overlap;
    a.k_1m = (float)(1 - mo); a.k_1p = (float)(1 + mo); a.k_m2 = (float)(-2 * mo); a.k_m1 = (float)(mo - 1);
    a.k_16 = (float)(4 * (4 * mo));
    hipStream_t s = as_stream(stream);
    ZeroArgs z{};
    const unsigned long long slots = (unsigned long long)num_tasks * batch * max_objs;
    z.p[0] = reinterpret_cast<unsigned char *>(heatmap); z.n[0] = (unsigned long long)batch * total * h * w * 4;
    z.p[1] = reinterpret_cast<unsigned char *>(anno_box); z.n[1] = slots * 40;
    z.p[2] = reinterpret_cast<unsigned char *>(ind); z.n[2] = slots * 8;
    z.p[3] = mask; z.n[3] = slots;
    zero_segments_kernel<<<1024, 256, 0, s>>>(z);
    if (int rc = check_launch("zero_segments_kernel")) return rc;
    if (n_max > 0) {
        targets_kernel<<<dim3(n_max, batch), 64, 0, s>>>(a);
        if (int rc = check_launch("targets_kernel")) return rc;
    }
    return SGV3D_OK;
}
