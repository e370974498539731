"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by sgv3d_amd/).

CPU restatement (numpy / plain Python, float64) of the KITTI-AP evaluator:

  evaluators/kitti_utils/rotate_iou.py:208-281   rotated-rectangle intersection -> rotated_overlap (here by
                                                 Sutherland-Hodgman clipping in float64: an independent algorithm, so
                                                 agreement with the float32 kernel is a real check, to ~1e-5)
  evaluators/kitti_utils/eval.py:120-160         d3_box_overlap                  -> d3_overlap
  evaluators/kitti_utils/eval.py:28-79           clean_data                      -> clean_data
  evaluators/kitti_utils/eval.py:157-277         compute_statistics_jit          -> frame_statistics
  evaluators/kitti_utils/eval.py:7-25            get_thresholds                  -> recall_thresholds
  evaluators/kitti_utils/eval.py:441-572         eval_class                      -> eval_class
  evaluators/kitti_utils/eval.py:574-584         get_mAP / get_mAP_R40           -> average_precision
  evaluators/kitti_utils/kitti_common.py:561-602 get_label_anno                  -> parse_label_text

Pinned by tests/golden/kitti_eval.npz (the reference's own modules executed in the build container, numba replaced by
an identity decorator; tests/golden/make_golden_aux.py)."""
import math

import numpy as np

CATEGORY_MAP = {'Car': 'Car', 'Bus': 'Car', 'Pedestrian': 'Pedestrian', 'Cyclist': 'Cyclist'}


def parse_label_text(text):
    rows = [ln.strip().split(' ') for ln in text.splitlines()] if len(text) >= 15 else []
    v = np.array([[float(x) for x in r[1:15]] for r in rows], np.float64).reshape(-1, 14)
    return dict(name=np.array([CATEGORY_MAP[r[0]] for r in rows]), truncated=v[:, 0], occluded=v[:, 1], alpha=v[:, 2],
                bbox=v[:, 3:7], dimensions=v[:, [9, 7, 8]], location=v[:, 10:13], rotation_y=v[:, 13],
                score=np.array([float(r[15]) for r in rows]) if rows and len(rows[0]) == 16 else np.zeros(len(rows)))


# ------------------------------------------------------------------------------------------------ overlaps
def _corners(b):
    x, y, dx, dy, a = (float(v) for v in b)
    c, s = math.cos(a), math.sin(a)
    return [(c * px + s * py + x, -s * px + c * py + y) for px, py in
            ((-dx / 2, -dy / 2), (-dx / 2, dy / 2), (dx / 2, dy / 2), (dx / 2, -dy / 2))]


def _clip(poly, a, b):
    """Part of ``poly`` on the left of (or on) the directed line a -> b."""
    out = []
    side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
    for i, p in enumerate(poly):
        q = poly[(i + 1) % len(poly)]
        sp, sq = side(p), side(q)
        if sp >= 0:
            out.append(p)
        if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_intersection(b1, b2):
    poly, clipper = _corners(b1), _corners(b2)
    orient = sum(clipper[i][0] * clipper[(i + 1) % 4][1] - clipper[(i + 1) % 4][0] * clipper[i][1] for i in range(4))
    if orient < 0:
        clipper = clipper[::-1]
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, clipper[i], clipper[(i + 1) % 4])
    return abs(sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1]
                   for i in range(len(poly)))) / 2 if len(poly) >= 3 else 0.0


def rotated_overlap(boxes, qboxes, criterion=-1):
    """[N, 5] x [K, 5] -> [N, K]; criterion 0 divides by the area of the QUERY box (rbox1 of devRotateIoUEval, :334-336)."""
    out = np.zeros((len(boxes), len(qboxes)))
    for i, b in enumerate(boxes):
        for j, q in enumerate(qboxes):
            inter = rotated_intersection(q, b)
            a1, a2 = q[2] * q[3], b[2] * b[3]
            out[i, j] = {-1: inter / (a1 + a2 - inter), 0: inter / a1, 1: inter / a2}.get(criterion, inter)
    return out


def d3_overlap(boxes, qboxes, criterion=-1):
    out = rotated_overlap(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2)
    for i, b in enumerate(boxes):
        for j, q in enumerate(qboxes):
            if out[i, j] > 0:
                iw = min(b[1], q[1]) - max(b[1] - b[4], q[1] - q[4])
                if iw > 0:
                    v1, v2, inc = b[3] * b[4] * b[5], q[3] * q[4] * q[5], iw * out[i, j]
                    out[i, j] = inc / {-1: v1 + v2 - inc, 0: v1, 1: v2}.get(criterion, inc)
                else:
                    out[i, j] = 0.0
    return out


def box2d_overlap(boxes, qboxes, criterion=-1):
    out = np.zeros((len(boxes), len(qboxes)))
    for i, b in enumerate(boxes):
        for j, q in enumerate(qboxes):
            iw = min(b[2], q[2]) - max(b[0], q[0])
            ih = min(b[3], q[3]) - max(b[1], q[1])
            if iw > 0 and ih > 0:
                ab, aq = (b[2] - b[0]) * (b[3] - b[1]), (q[2] - q[0]) * (q[3] - q[1])
                out[i, j] = iw * ih / {-1: ab + aq - iw * ih, 0: ab, 1: aq}.get(criterion, 1.0)
    return out


# ------------------------------------------------------------------------------------------------ statistics
def clean_data(gt, dt, current_class, difficulty):
    cls = ['car', 'pedestrian', 'cyclist', 'bus'][current_class]
    ign_gt, ign_dt, dc, valid = [], [], [], 0
    for i, name in enumerate(gt['name']):
        n = str(name).lower()
        kind = 1 if n == cls else (0 if (cls, n) in (('pedestrian', 'person_sitting'), ('car', 'van')) else -1)
        h = gt['bbox'][i][3] - gt['bbox'][i][1]
        hard = gt['occluded'][i] > [0, 1, 2][difficulty] or gt['truncated'][i] > [0.15, 0.3, 0.5][difficulty] or h <= [40, 25, 25][difficulty]
        if kind == 1 and not hard:
            ign_gt.append(0)
            valid += 1
        elif kind == 0 or (hard and kind == 1):
            ign_gt.append(1)
        else:
            ign_gt.append(-1)
        if name == 'DontCare':
            dc.append(gt['bbox'][i])
    for i, name in enumerate(dt['name']):
        h = abs(dt['bbox'][i][3] - dt['bbox'][i][1])
        ign_dt.append(1 if h < [40, 25, 25][difficulty] else (0 if str(name).lower() == cls else -1))
    return valid, np.array(ign_gt, np.int64), np.array(ign_dt, np.int64), np.array(dc, np.float64).reshape(-1, 4)


def frame_statistics(overlaps, gt_datas, dt_datas, ign_gt, ign_dt, dc, metric, min_overlap, thresh=0.0, compute_fp=False,
                     compute_aos=False):
    """-> (tp, fp, fn, similarity, scores of the true positives); overlaps [detections, ground truth]."""
    D, G = len(dt_datas), len(gt_datas)
    scores = dt_datas[:, 5] if D else np.zeros(0)
    taken = [False] * D
    low = [bool(compute_fp and scores[j] < thresh) for j in range(D)]
    NONE = -10000000
    tp = fp = fn = 0
    similarity = 0
    tps, delta = [], []
    for i in range(G):
        if ign_gt[i] == -1:
            continue
        det, valid, best, from_ignored = -1, NONE, 0, False
        for j in range(D):
            if ign_dt[j] == -1 or taken[j] or low[j]:
                continue
            ov = overlaps[j, i]
            if not compute_fp and ov > min_overlap and scores[j] > valid:
                det, valid = j, scores[j]
            elif compute_fp and ov > min_overlap and (ov > best or from_ignored) and ign_dt[j] == 0:
                best, det, valid, from_ignored = ov, j, 1, False
            elif compute_fp and ov > min_overlap and valid == NONE and ign_dt[j] == 1:
                det, valid, from_ignored = j, 1, True
        if valid == NONE and ign_gt[i] == 0:
            fn += 1
        elif valid != NONE and (ign_gt[i] == 1 or ign_dt[det] == 1):
            taken[det] = True
        elif valid != NONE:
            tp += 1
            tps.append(scores[det])
            if compute_aos:
                delta.append(gt_datas[i, 4] - dt_datas[det, 4])
            taken[det] = True
    if compute_fp:
        fp = sum(1 for j in range(D) if not (taken[j] or ign_dt[j] in (-1, 1) or low[j]))
        if metric == 0 and len(dc):
            ov = box2d_overlap(dt_datas[:, :4], dc, 0)
            for k in range(len(dc)):
                for j in range(D):
                    if taken[j] or ign_dt[j] in (-1, 1) or low[j]:
                        continue
                    if ov[j, k] > min_overlap:
                        taken[j] = True
                        fp -= 1
        if compute_aos:
            similarity = sum((1.0 + math.cos(d)) / 2.0 for d in delta) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.array(tps)


def recall_thresholds(scores, num_gt, num_sample_pts=41):
    scores = np.sort(scores)[::-1]
    out, current = [], 0.0
    for i, s in enumerate(scores):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < len(scores) - 1 else left
        if (right - current) < (current - left) and i < len(scores) - 1:
            continue
        out.append(s)
        current += 1 / (num_sample_pts - 1.0)
    return out


def frame_overlaps(gt, dt, metric):
    if metric == 0:
        return box2d_overlap(dt['bbox'], gt['bbox'])
    if metric == 1:
        f = lambda a: np.concatenate([a['location'][:, [0, 2]], a['dimensions'][:, [0, 2]], a['rotation_y'][:, None]], 1)
        return rotated_overlap(f(dt), f(gt)).astype(np.float32).astype(np.float64)
    f = lambda a: np.concatenate([a['location'], a['dimensions'], a['rotation_y'][:, None]], 1)
    return d3_overlap(f(dt), f(gt)).astype(np.float32).astype(np.float64)


def eval_class(gt_annos, dt_annos, classes, difficulties, metric, min_overlaps, compute_aos=False):
    ov = [frame_overlaps(g, d, metric) for g, d in zip(gt_annos, dt_annos)]
    gtd = [np.concatenate([g['bbox'].reshape(-1, 4), g['alpha'].reshape(-1, 1)], 1) for g in gt_annos]
    dtd = [np.concatenate([d['bbox'].reshape(-1, 4), d['alpha'].reshape(-1, 1), d['score'].reshape(-1, 1)], 1) for d in dt_annos]
    shape = [len(classes), len(difficulties), len(min_overlaps), 41]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(classes):
        for l, diff in enumerate(difficulties):
            cl = [clean_data(g, d, cls, diff) for g, d in zip(gt_annos, dt_annos)]
            total_valid = sum(c[0] for c in cl)
            for k, mo in enumerate(min_overlaps[:, metric, m]):
                tps = [frame_statistics(ov[i], gtd[i], dtd[i], cl[i][1], cl[i][2], cl[i][3], metric, mo)[4] for i in range(len(ov))]
                thr = recall_thresholds(np.concatenate(tps) if tps else np.zeros(0), total_valid)
                pr = np.zeros((len(thr), 4))
                for i in range(len(ov)):
                    for t, th in enumerate(thr):
                        tp, fp, fn, sim, _ = frame_statistics(ov[i], gtd[i], dtd[i], cl[i][1], cl[i][2], cl[i][3], metric, mo, th,
                                                              True, compute_aos)
                        pr[t, :3] += (tp, fp, fn)
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(divide='ignore', invalid='ignore'):
                    for t in range(len(thr)):
                        recall[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 2])
                        precision[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 1])
                        if compute_aos:
                            aos[m, l, k, t] = pr[t, 3] / (pr[t, 0] + pr[t, 1])
                for t in range(len(thr)):
                    precision[m, l, k, t] = np.max(precision[m, l, k, t:])
                    recall[m, l, k, t] = np.max(recall[m, l, k, t:])
                    if compute_aos:
                        aos[m, l, k, t] = np.max(aos[m, l, k, t:])
    return dict(precision=precision, recall=recall, orientation=aos)


def average_precision(prec, metric="R40"):
    if metric == "R40":
        return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100
