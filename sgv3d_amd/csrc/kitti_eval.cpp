// Host side of the KITTI-AP evaluator: the matching / precision-recall sweep that the
// reference runs through numba-jitted functions on the CPU -- get_thresholds (evaluators/kitti_utils/eval.py:7-25),
// image_box_overlap (:82-111), compute_statistics_jit (:157-277), fused_compute_statistics (:289-335) and the loop of
// eval_class around them (:487-556).  Plain C++ on host pointers; the overlaps come from sgv3d_rotate_iou_pairs.
//
// One call handles one (class, difficulty, minimum overlap) cell of the result table.  The threshold sweep is spread
// over host threads BY THRESHOLD, every thread walking the images in order, so the sums (including the float64
// orientation similarity) are formed in the same order on every run and for every thread count.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

struct Frame {                    // one image's slice of the flattened inputs
    int n_gt, n_dt, n_dc;
    const double *overlap;        // [n_dt][n_gt]
    const double *gt;             // [n_gt][5]: 2-D box, alpha
    const double *dt;             // [n_dt][6]: 2-D box, alpha, score
    const int64_t *ign_gt, *ign_dt;
    const double *dc;             // [n_dc][4]
};

struct Counts { long long tp = 0, fp = 0, fn = 0; double similarity = 0; bool has_similarity = false; };

// compute_statistics_jit: greedy assignment of detections to ground-truth boxes of one image.
// `tp_scores` (may be null) receives the score of every true positive (the compute_fp = False pass).
Counts match_frame(const Frame &f, int metric, double min_overlap, double thresh, bool compute_fp, bool compute_aos,
                   std::vector<double> *tp_scores) {
    const int G = f.n_gt, D = f.n_dt;
    std::vector<char> assigned(D, 0), below(D, 0);
    if (compute_fp)
        for (int j = 0; j < D; ++j) below[j] = f.dt[j * 6 + 5] < thresh;
    constexpr double kNone = -10000000.0;
    Counts c;
    std::vector<double> delta;
    for (int i = 0; i < G; ++i) {
        if (f.ign_gt[i] == -1) continue;
        int det = -1;
        double valid = kNone, max_overlap = 0;
        bool assigned_ignored = false;
        for (int j = 0; j < D; ++j) {
            if (f.ign_dt[j] == -1 || assigned[j] || below[j]) continue;
            const double ov = f.overlap[(size_t)j * G + i];
            const double score = f.dt[j * 6 + 5];
            if (!compute_fp && ov > min_overlap && score > valid) {
                det = j;
                valid = score;
            } else if (compute_fp && ov > min_overlap && (ov > max_overlap || assigned_ignored) && f.ign_dt[j] == 0) {
                max_overlap = ov;
                det = j;
                valid = 1;
                assigned_ignored = false;
            } else if (compute_fp && ov > min_overlap && valid == kNone && f.ign_dt[j] == 1) {
                det = j;
                valid = 1;
                assigned_ignored = true;
            }
        }
        if (valid == kNone && f.ign_gt[i] == 0) {
            ++c.fn;
        } else if (valid != kNone && (f.ign_gt[i] == 1 || f.ign_dt[det] == 1)) {
            assigned[det] = 1;
        } else if (valid != kNone) {
            ++c.tp;
            if (tp_scores) tp_scores->push_back(f.dt[det * 6 + 5]);
            if (compute_aos) delta.push_back(f.gt[i * 5 + 4] - f.dt[det * 6 + 4]);
            assigned[det] = 1;
        }
    }
    if (compute_fp) {
        for (int j = 0; j < D; ++j)
            if (!(assigned[j] || f.ign_dt[j] == -1 || f.ign_dt[j] == 1 || below[j])) ++c.fp;
        long long nstuff = 0;
        if (metric == 0) {
            // detections that fall on a DontCare region are not false positives: overlap = intersection / detection area
            for (int k = 0; k < f.n_dc; ++k) {
                const double *q = f.dc + k * 4;
                for (int j = 0; j < D; ++j) {
                    if (assigned[j] || f.ign_dt[j] == -1 || f.ign_dt[j] == 1 || below[j]) continue;
                    const double *b = f.dt + j * 6;
                    double ov = 0;
                    const double iw = std::min(b[2], q[2]) - std::max(b[0], q[0]);
                    if (iw > 0) {
                        const double ih = std::min(b[3], q[3]) - std::max(b[1], q[1]);
                        if (ih > 0) ov = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
                    }
                    if (ov > min_overlap) {
                        assigned[j] = 1;
                        ++nstuff;
                    }
                }
            }
        }
        c.fp -= nstuff;
        if (compute_aos) {
            c.has_similarity = c.tp > 0 || c.fp > 0;
            double s = 0;
            for (double d : delta) s += (1.0 + std::cos(d)) / 2.0;
            c.similarity = s;
        }
    }
    return c;
}

// get_thresholds: scores at which the recall crosses the 41 sample points
std::vector<double> recall_thresholds(std::vector<double> scores, long long num_gt, int num_sample_pts = 41) {
    std::sort(scores.begin(), scores.end(), std::greater<double>());
    std::vector<double> out;
    double current = 0;
    const size_t n = scores.size();
    for (size_t i = 0; i < n; ++i) {
        const double l = (double)(i + 1) / (double)num_gt;
        const double r = i + 1 < n ? (double)(i + 2) / (double)num_gt : l;
        if ((r - current) < (current - l) && i + 1 < n) continue;
        out.push_back(scores[i]);
        current += 1.0 / (num_sample_pts - 1.0);
    }
    return out;
}

double nanmax(const double *v, int n) {        // np.max: a NaN anywhere gives NaN
    double m = v[0];
    for (int i = 0; i < n; ++i) {
        if (std::isnan(v[i])) return std::numeric_limits<double>::quiet_NaN();
        m = std::max(m, v[i]);
    }
    return m;
}

}  // namespace

extern "C" int sgv3d_kitti_eval_curves(int num_images, const int32_t *gt_num, const int32_t *dt_num, const int32_t *dc_num,
                                       const double *overlaps, const double *gt_datas, const double *dt_datas,
                                       const int64_t *ignored_gt, const int64_t *ignored_det, const double *dontcares,
                                       int metric, double min_overlap, int compute_aos, long long num_valid_gt,
                                       int num_threads, double *precision, double *recall, double *orientation,
                                       int *num_thresholds) {
    SGV3D_REQUIRE(num_images >= 0 && metric >= 0 && metric <= 2, "kitti_eval_curves: bad arguments");
    SGV3D_REQUIRE(precision && recall && orientation, "kitti_eval_curves: null output");
    SGV3D_REQUIRE(num_images == 0 || (gt_num && dt_num && dc_num), "kitti_eval_curves: null count arrays");
    constexpr int kPts = 41;
    std::vector<Frame> frames(num_images);
    size_t o = 0, g = 0, d = 0, c = 0;
    for (int m = 0; m < num_images; ++m) {
        SGV3D_REQUIRE(gt_num[m] >= 0 && dt_num[m] >= 0 && dc_num[m] >= 0, "kitti_eval_curves: negative count");
        Frame &f = frames[m];
        f.n_gt = gt_num[m]; f.n_dt = dt_num[m]; f.n_dc = dc_num[m];
        f.overlap = overlaps + o; f.gt = gt_datas + g * 5; f.dt = dt_datas + d * 6;
        f.ign_gt = ignored_gt + g; f.ign_dt = ignored_det + d; f.dc = dontcares + c * 4;
        o += (size_t)f.n_gt * f.n_dt; g += f.n_gt; d += f.n_dt; c += f.n_dc;
    }
    // pass 1: scores of the true positives at threshold 0 -> the recall sample thresholds (eval.py:493-509)
    std::vector<double> scores;
    for (const Frame &f : frames) match_frame(f, metric, min_overlap, 0.0, false, false, &scores);
    const std::vector<double> thr = recall_thresholds(scores, num_valid_gt);
    const int T = (int)thr.size();
    SGV3D_REQUIRE(T <= kPts, "kitti_eval_curves: more than 41 thresholds");
    if (num_thresholds) *num_thresholds = T;
    // pass 2: tp / fp / fn / similarity per threshold (fused_compute_statistics, eval.py:289-335)
    std::vector<Counts> pr(T);
    std::vector<double> sim(T, 0.0);
    auto sweep = [&](int t0, int t1) {
        for (int t = t0; t < t1; ++t)
            for (const Frame &f : frames) {
                const Counts r = match_frame(f, metric, min_overlap, thr[t], true, compute_aos != 0, nullptr);
                pr[t].tp += r.tp; pr[t].fp += r.fp; pr[t].fn += r.fn;
                if (compute_aos && r.has_similarity) sim[t] += r.similarity;
            }
    };
    int nt = std::max(1, std::min(num_threads, T));
    if (nt <= 1) {
        sweep(0, T);
    } else {
        std::vector<std::thread> pool;
        for (int w = 0; w < nt; ++w) pool.emplace_back(sweep, (int)((long long)T * w / nt), (int)((long long)T * (w + 1) / nt));
        for (auto &th : pool) th.join();
    }
    for (int i = 0; i < kPts; ++i) precision[i] = recall[i] = orientation[i] = 0.0;
    for (int t = 0; t < T; ++t) {                                         // eval.py:543-548 (0 / 0 = NaN, as numpy)
        const double tp = (double)pr[t].tp, fp = (double)pr[t].fp, fn = (double)pr[t].fn;
        recall[t] = tp / (tp + fn);
        precision[t] = tp / (tp + fp);
        if (compute_aos) orientation[t] = sim[t] / (tp + fp);
    }
    for (int t = 0; t < T; ++t) {                                         // eval.py:549-556: running maximum from the right
        precision[t] = nanmax(precision + t, kPts - t);
        recall[t] = nanmax(recall + t, kPts - t);
        if (compute_aos) orientation[t] = nanmax(orientation + t, kPts - t);
    }
    return SGV3D_OK;
}
