// Rotated-box overlaps of the KITTI-AP evaluator on the MI355X.
//
// Reference: the numba.cuda kernel rotate_iou_kernel_eval + devRotateIoUEval (evaluators/kitti_utils/rotate_iou.py:
// 17-337, launcher :340-378) and the 3-D overlap built on it (d3_box_overlap / d3_box_overlap_kernel,
// evaluators/kitti_utils/eval.py:120-160).  The reference evaluates a dense N x K matrix per "part" of ~15 images
// (every detection against every ground-truth box of the part) and keeps only the per-image diagonal blocks
// (eval.py:353-438); here ONE launch covers the whole validation set and only same-image pairs are evaluated:
//
//   out[out_off[m] + i * K_m + j] = overlap(box i of image m, query box j of image m),   i < N_m, j < K_m
//
// with the polygon clipping done per pair in float32 in the reference's operation order (this file is compiled with
// -ffp-contract=off: the inside / crossing tests of touching boxes depend on single roundings).  One workgroup = a
// 16 x 16 tile of pairs of one image; the eight corner coordinates of its 16 + 16 boxes are computed once into LDS.
// The dense call of the reference's signature is the one-image case.
#include "common.hpp"

using namespace sgv3d;

namespace {

constexpr int kTile = 16;

struct Quad { float c[8]; };

// rbbox_to_corners (rotate_iou.py:208-231): clockwise corners of (cx, cy, dx, dy, angle), rotated clockwise
__device__ __forceinline__ void corners_of(const double *b5, Quad &q) {
    const float cx = (float)b5[0], cy = (float)b5[1], xd = (float)b5[2], yd = (float)b5[3], ang = (float)b5[4];
    const float ac = (float)cos((double)ang), as = (float)sin((double)ang);
    const float px[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    const float py[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
    for (int i = 0; i < 4; ++i) {
        q.c[2 * i] = ac * px[i] + as * py[i] + cx;
        q.c[2 * i + 1] = -as * px[i] + ac * py[i] + cy;
    }
}

// point_in_quadrilateral (:170-186)
__device__ __forceinline__ bool inside(float x, float y, const float *c) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = x - c[0], ap1 = y - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1;
    const float abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1;
    const float adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

// line_segment_intersection (:73-118): edge i of p1 against edge j of p2
__device__ __forceinline__ bool crossing(const float *p1, const float *p2, int i, int j, float &ox, float &oy) {
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    const float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    const float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1;
    const float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd == bcd) return false;
    const bool abc = CA1 * BA0 > BA1 * CA0;
    const bool abd = DA1 * BA0 > BA1 * DA0;
    if (abc == abd) return false;
    const float DC0 = D0 - C0, DC1 = D1 - C1;
    const float ABBA = A0 * B1 - B0 * A1;
    const float CDDC = C0 * D1 - D0 * C1;
    const float DH = BA1 * DC0 - BA0 * DC1;
    ox = (ABBA * DC0 - BA0 * CDDC) / DH;
    oy = (ABBA * DC1 - BA1 * CDDC) / DH;
    return true;
}

// inter (:234-256): area of the intersection polygon of two rotated rectangles
__device__ float intersection_area(const float *c1, const float *c2) {
    float pts[16];
    int n = 0;
    auto push = [&](float x, float y) {
        if (n < 8) { pts[2 * n] = x; pts[2 * n + 1] = y; }     // the reference's buffer holds 8 points
        ++n;
    };
    for (int i = 0; i < 4; ++i) {                              // quadrilateral_intersection (:189-205)
        if (inside(c1[2 * i], c1[2 * i + 1], c2)) push(c1[2 * i], c1[2 * i + 1]);
        if (inside(c2[2 * i], c2[2 * i + 1], c1)) push(c2[2 * i], c2[2 * i + 1]);
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float x, y;
            if (crossing(c1, c2, i, j, x, y)) push(x, y);
        }
    if (n > 8) n = 8;
    if (n > 0) {                                               // sort_vertex_in_convex_polygon (:33-69)
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
        cx /= (float)n; cy /= (float)n;
        float vs[8];
        for (int i = 0; i < n; ++i) {
            float v0 = pts[2 * i] - cx, v1 = pts[2 * i + 1] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {                          // insertion sort on the pseudo-angle
            if (vs[i - 1] > vs[i]) {
                const float t = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > t) {
                    vs[j] = vs[j - 1];
                    pts[2 * j] = pts[2 * j - 2];
                    pts[2 * j + 1] = pts[2 * j - 1];
                    --j;
                }
                vs[j] = t; pts[2 * j] = tx; pts[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0;                                         // area (:23-30): fan of triangles from point 0
    for (int i = 0; i < n - 2; ++i) {
        const float *a = pts, *b = pts + 2 * i + 2, *c = pts + 2 * i + 4;
        const float cr = (a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0]);
        area += fabs((double)cr / 2.0);
    }
    return (float)area;
}

struct IouArgs {
    const int *box_off, *qbox_off, *tile_off;    // [images + 1] each (device)
    const long long *out_off;                    // [images] (device)
    const double *boxes, *qboxes;                // [*, dim]
    float *out;
    int images, dim, criterion, mode3d;
};

__global__ void __launch_bounds__(kTile * kTile) rotate_iou_kernel(IouArgs a) {
    __shared__ Quad s_b[kTile], s_q[kTile];
    __shared__ int s_img;
    if (threadIdx.x == 0) {                      // image of this tile: last m with tile_off[m] <= blockIdx.x
        int lo = 0, hi = a.images - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.tile_off[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        s_img = lo;
    }
    __syncthreads();
    const int m = s_img;
    const int n = a.box_off[m + 1] - a.box_off[m], k = a.qbox_off[m + 1] - a.qbox_off[m];
    const int tiles_k = (k + kTile - 1) / kTile;
    const int t = (int)blockIdx.x - a.tile_off[m];
    const int i0 = (t / tiles_k) * kTile, j0 = (t % tiles_k) * kTile;
    const int ti = threadIdx.x / kTile, tj = threadIdx.x % kTile;
    const int sel5[5] = {0, 2, 3, 5, 6};         // eval.py:155-156: (x, z, l, w, ry) of a camera-frame box
    if (threadIdx.x < 2 * kTile) {
        const bool isq = threadIdx.x >= kTile;
        const int r = threadIdx.x % kTile;
        const int idx = (isq ? j0 : i0) + r;
        if (idx < (isq ? k : n)) {
            const double *src = (isq ? a.qboxes + (size_t)(a.qbox_off[m] + idx) * a.dim : a.boxes + (size_t)(a.box_off[m] + idx) * a.dim);
            double b5[5];
            for (int c = 0; c < 5; ++c) b5[c] = a.mode3d ? src[sel5[c]] : src[c];
            corners_of(b5, isq ? s_q[r] : s_b[r]);
        }
    }
    __syncthreads();
    const int i = i0 + ti, j = j0 + tj;
    if (i >= n || j >= k) return;
    const double *bb = a.boxes + (size_t)(a.box_off[m] + i) * a.dim;
    const double *qq = a.qboxes + (size_t)(a.qbox_off[m] + j) * a.dim;
    // devRotateIoUEval(rbox1 = query box, rbox2 = box) (:259-281, call at :334-336)
    const float inter = intersection_area(s_q[tj].c, s_b[ti].c);
    float res;
    if (!a.mode3d) {
        const float area1 = (float)qq[2] * (float)qq[3], area2 = (float)bb[2] * (float)bb[3];
        if (a.criterion == -1) res = inter / (area1 + area2 - inter);
        else if (a.criterion == 0) res = inter / area1;
        else if (a.criterion == 1) res = inter / area2;
        else res = inter;
    } else {
        // d3_box_overlap_kernel (eval.py:120-150) in float64 on the float32 BEV intersection; y points down, a box spans
        // [y - h, y] with h = dimension 4
        res = 0.f;
        if (inter > 0.f) {
            const double iw = fmin(bb[1], qq[1]) - fmax(bb[1] - bb[4], qq[1] - qq[4]);
            if (iw > 0) {
                const double v1 = bb[3] * bb[4] * bb[5], v2 = qq[3] * qq[4] * qq[5];
                const double inc = iw * (double)inter;
                double ua;
                if (a.criterion == -1) ua = v1 + v2 - inc;
                else if (a.criterion == 0) ua = v1;
                else if (a.criterion == 1) ua = v2;
                else ua = inc;
                res = (float)(inc / ua);
            }
        }
    }
    a.out[a.out_off[m] + (long long)i * k + j] = res;
}

}  // namespace

extern "C" int sgv3d_rotate_iou_pairs(int num_images, int num_tiles, const int32_t *box_offsets, const int32_t *qbox_offsets,
                                      const int32_t *tile_offsets, const long long *out_offsets, const double *boxes,
                                      const double *qboxes, int box_dim, int criterion, float *out, void *stream) {
    SGV3D_REQUIRE(num_images > 0 && num_tiles >= 0, "rotate_iou_pairs: bad sizes");
    SGV3D_REQUIRE(box_dim == 5 || box_dim == 7, "rotate_iou_pairs: box_dim is 5 (BEV: x, y, dx, dy, angle) or 7 (camera-frame 3-D box)");
    SGV3D_REQUIRE(criterion >= -1 && criterion <= 2, "rotate_iou_pairs: criterion is -1 (IoU), 0, 1 or 2 (intersection)");
    if (num_tiles == 0) return SGV3D_OK;
    SGV3D_REQUIRE(box_offsets && qbox_offsets && tile_offsets && out_offsets && boxes && qboxes && out, "rotate_iou_pairs: null pointer");
    IouArgs a{};
    a.box_off = box_offsets; a.qbox_off = qbox_offsets; a.tile_off = tile_offsets; a.out_off = out_offsets;
    a.boxes = boxes; a.qboxes = qboxes; a.out = out;
    a.images = num_images; a.dim = box_dim; a.criterion = criterion; a.mode3d = box_dim == 7;
    rotate_iou_kernel<<<num_tiles, kTile * kTile, 0, as_stream(stream)>>>(a);
    return check_launch("rotate_iou_kernel");
}
