"""KITTI average precision (2-D box, BEV, 3-D, orientation) of detection results -- ``kitti_eval`` of
evaluators/kitti_utils/eval.py:651-781 with the same inputs (annotation dicts of ``kitti_common.get_label_annos``),
result text and result keys.

What runs where (the reference: numpy + numba-jitted CPU loops + a numba.cuda kernel per "part" of images):

* overlaps (``calculate_overlaps``, replaces calculate_iou_partly :338-438): 2-D boxes in numpy; BEV and 3-D through
  ONE launch of the rotated-box kernel over all same-image pairs of the set (csrc/rotate_iou.hip);
* ``clean_data`` (:28-79): class / difficulty filtering, host Python (string logic);
* matching, recall thresholds, precision / recall / orientation curves: ``sgv3d_kitti_eval_curves`` (csrc/kitti_eval.cpp,
  host C++, the numba-jitted part of the reference), one call per (class, difficulty, minimum overlap).

There is no CPU path for the overlaps: without the HIP library / a GPU the BEV and 3-D metrics raise.
"""
import ctypes
import os

import numpy as np

from ... import _lib
from .rotate_iou import rotate_iou_pairs

__all__ = ['kitti_eval', 'eval_class', 'clean_data', 'calculate_overlaps', 'image_box_overlap', 'bev_box_overlap',
           'd3_box_overlap', 'get_mAP', 'get_mAP_R40', 'do_eval']

_CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'bus']          # clean_data, :29
_MIN_HEIGHT = [40, 25, 25]
_MAX_OCCLUSION = [0, 1, 2]
_MAX_TRUNCATION = [0.15, 0.3, 0.5]
_NEIGHBOUR = {'pedestrian': 'person_sitting', 'car': 'van'}    # counted neither as hit nor as miss (:44-48)


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """-> (num_valid_gt, ignored_gt int64 [G], ignored_dt int64 [D], dc_bboxes float64 [n, 4]); flags: 0 evaluate,
    1 ignore (neighbouring class or beyond the difficulty limits), -1 other class."""
    cls = _CLASS_NAMES[current_class]
    names = np.char.lower(np.asarray(gt_anno['name'], dtype=str)) if len(gt_anno['name']) else np.zeros(0, dtype=str)
    bbox = np.asarray(gt_anno['bbox'], np.float64).reshape(-1, 4)
    same = names == cls
    near = names == _NEIGHBOUR.get(cls, '\0') if len(names) else np.zeros(0, bool)
    hard = ((np.asarray(gt_anno['occluded']) > _MAX_OCCLUSION[difficulty]) |
            (np.asarray(gt_anno['truncated']) > _MAX_TRUNCATION[difficulty]) |
            ((bbox[:, 3] - bbox[:, 1]) <= _MIN_HEIGHT[difficulty]))
    ignored_gt = np.full(len(names), -1, np.int64)
    ignored_gt[near | (same & hard)] = 1
    ignored_gt[same & ~hard] = 0
    dc = bbox[np.asarray(gt_anno['name'], dtype=str) == 'DontCare'] if len(names) else np.zeros((0, 4))
    dnames = np.char.lower(np.asarray(dt_anno['name'], dtype=str)) if len(dt_anno['name']) else np.zeros(0, dtype=str)
    dbox = np.asarray(dt_anno['bbox'], np.float64).reshape(-1, 4)
    ignored_dt = np.where(dnames == cls, 0, -1).astype(np.int64) if len(dnames) else np.zeros(0, np.int64)
    ignored_dt[np.abs(dbox[:, 3] - dbox[:, 1]) < _MIN_HEIGHT[difficulty]] = 1
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc.reshape(-1, 4).astype(np.float64)


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """Axis-aligned 2-D overlaps [N, K] (:82-111): criterion -1 IoU, 0 / area of ``boxes``, 1 / area of ``query_boxes``."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)[:, None, :]
    q = np.asarray(query_boxes, np.float64).reshape(-1, 4)[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])
    ab = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    aq = (q[..., 2] - q[..., 0]) * (q[..., 3] - q[..., 1])
    inter = iw * ih
    ua = {-1: ab + aq - inter, 0: ab + 0 * aq, 1: aq + 0 * ab}.get(criterion, np.ones_like(inter))
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where((iw > 0) & (ih > 0), inter / ua, 0.0)


def bev_box_overlap(boxes, qboxes, criterion=-1):
    """[N, 5] x [K, 5] BEV rectangles (x, y, dx, dy, angle) -> [N, K] (:114-117)."""
    return rotate_iou_pairs([np.asarray(boxes)], [np.asarray(qboxes)], criterion)[0]


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """[N, 7] x [K, 7] camera-frame boxes (x, y, z, l, h, w, ry) -> [N, K] 3-D overlaps (:153-160)."""
    return rotate_iou_pairs([np.asarray(boxes)], [np.asarray(qboxes)], criterion)[0]


def _rboxes(anno, metric):
    loc = np.asarray(anno['location'], np.float64).reshape(-1, 3)
    dims = np.asarray(anno['dimensions'], np.float64).reshape(-1, 3)
    rots = np.asarray(anno['rotation_y'], np.float64).reshape(-1, 1)
    if metric == 1:
        return np.concatenate([loc[:, [0, 2]], dims[:, [0, 2]], rots], 1)           # :364-377
    return np.concatenate([loc, dims, rots], 1)                                     # :381-390


def calculate_overlaps(gt_annos, dt_annos, metric):
    """Per image the float64 matrix [detections, ground truth] (eval_class passes the two lists to calculate_iou_partly in
    this order, :470).  metric 0: 2-D boxes, 1: BEV, 2: 3-D."""
    assert len(gt_annos) == len(dt_annos)
    if metric == 0:
        return [image_box_overlap(d['bbox'], g['bbox']) for g, d in zip(gt_annos, dt_annos)]
    if metric not in (1, 2):
        raise ValueError('unknown metric')
    out = rotate_iou_pairs([_rboxes(d, metric) for d in dt_annos], [_rboxes(g, metric) for g in gt_annos], -1)
    return [o.astype(np.float64) for o in out]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_threads=None):
    """precision / recall / orientation arrays [class, difficulty, min_overlap, 41] (:441-572)."""
    assert len(gt_annos) == len(dt_annos)
    overlaps = calculate_overlaps(gt_annos, dt_annos, metric)
    M = len(gt_annos)
    num_threads = num_threads or min(16, os.cpu_count() or 1)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), 41]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    if M == 0:
        return {'recall': recall, 'precision': precision, 'orientation': aos}
    lib = _lib.load()
    flat_ov = np.ascontiguousarray(np.concatenate([o.reshape(-1) for o in overlaps]) if M else np.zeros(0), np.float64)
    gt_num = np.array([len(g['name']) for g in gt_annos], np.int32)
    dt_num = np.array([len(d['name']) for d in dt_annos], np.int32)
    gt_datas = np.ascontiguousarray(np.concatenate(
        [np.concatenate([np.asarray(g['bbox'], np.float64).reshape(-1, 4), np.asarray(g['alpha'], np.float64).reshape(-1, 1)], 1)
         for g in gt_annos], 0))
    dt_datas = np.ascontiguousarray(np.concatenate(
        [np.concatenate([np.asarray(d['bbox'], np.float64).reshape(-1, 4), np.asarray(d['alpha'], np.float64).reshape(-1, 1),
                         np.asarray(d['score'], np.float64).reshape(-1, 1)], 1) for d in dt_annos], 0))
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            cleaned = [clean_data(g, d, current_class, difficulty) for g, d in zip(gt_annos, dt_annos)]
            num_valid = sum(c[0] for c in cleaned)
            ign_gt = np.ascontiguousarray(np.concatenate([c[1] for c in cleaned]), np.int64)
            ign_dt = np.ascontiguousarray(np.concatenate([c[2] for c in cleaned]), np.int64)
            dcs = np.ascontiguousarray(np.concatenate([c[3] for c in cleaned], 0), np.float64)
            dc_num = np.array([len(c[3]) for c in cleaned], np.int32)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                p, r, o = precision[m, l, k], recall[m, l, k], aos[m, l, k]
                rc = lib.sgv3d_kitti_eval_curves(
                    M, _ptr(gt_num), _ptr(dt_num), _ptr(dc_num), _ptr(flat_ov), _ptr(gt_datas), _ptr(dt_datas), _ptr(ign_gt),
                    _ptr(ign_dt), _ptr(dcs), int(metric), float(min_overlap), 1 if compute_aos else 0, int(num_valid),
                    int(num_threads), _ptr(p), _ptr(r), _ptr(o), None)
                _lib.check(rc, "sgv3d_kitti_eval_curves")
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def get_mAP(prec):
    """11-point interpolated AP in percent (:574-578)."""
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100


def get_mAP_R40(prec):
    """40-point AP in percent (:580-584)."""
    return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types=('bbox', 'bev', '3d'), metric="R40"):
    """-> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos), arrays [class, difficulty, min_overlap] or None (:596-638)."""
    difficultys = [0, 1, 2]
    ap = get_mAP_R40 if metric == 'R40' else get_mAP
    mAP_bbox = mAP_aos = mAP_bev = mAP_3d = None
    if 'bbox' in eval_types:
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos=('aos' in eval_types))
        mAP_bbox = ap(ret['precision'])
        if 'aos' in eval_types:
            mAP_aos = ap(ret['orientation'])
    if 'bev' in eval_types:
        mAP_bev = ap(eval_class(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps)['precision'])
    if '3d' in eval_types:
        mAP_3d = ap(eval_class(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps)['precision'])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


_CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Bus', 4: 'Person_sitting'}


class _Metric:
    """One line kind of the report: its label in the text, its key in the result dictionary (None: text only), the AP array
    [class, difficulty, strict | loose] and the number of digits it is printed with."""
    __slots__ = ('label', 'key', 'values', 'digits')

    def __init__(self, label, key, values, digits):
        self.label, self.key, self.values, self.digits = label, key, values, digits


def _row(head, triple, digits):
    return head + ', '.join(f'{float(v):.{digits}f}' for v in triple)


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=('bbox', 'bev', '3d'), metric="R40"):
    """-> (result text, dict of 'KITTI/<class>_<3D|BEV|2D>_<difficulty>_<strict|loose>' and 'KITTI/Overall_*' values)."""
    eval_types = list(eval_types)
    assert len(eval_types) > 0, 'must contain at least one evaluation type'
    if 'aos' in eval_types:
        assert 'bbox' in eval_types, 'must evaluate bbox when evaluating aos'
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3)                                                   # :672-674
    loose = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])   # :675-677
    min_overlaps = np.stack([strict, loose], axis=0)                       # [2, metric, class]
    name_to_class = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    # orientation is evaluated when the detections carry alpha and the ground truth's is valid (:700-712; the reference
    # indexes the first ground-truth file's first object and fails on an empty first file -- empty files are skipped here)
    pred_alpha = any(a['alpha'].shape[0] != 0 for a in dt_annos)
    first_gt = next((a for a in gt_annos if a['alpha'].shape[0] != 0), None)
    compute_aos = pred_alpha and first_gt is not None and first_gt['alpha'][0] != -10
    if compute_aos and 'aos' not in eval_types:
        eval_types.append('aos')
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types, metric=metric)
    # The report is the wire format the AP text is compared on (evaluators/kitti_utils/eval.py:722-781): one table says which
    # metrics exist, in which order, under which label and dictionary key, with how many digits
    table = [m for m in (_Metric('bbox', '2D', mAPbbox, 4), _Metric('bev ', 'BEV', mAPbev, 4), _Metric('3d  ', '3D', mAP3d, 4),
                         _Metric('aos ', None, mAPaos if compute_aos else None, 2)) if m.values is not None]
    difficulty = ('easy', 'moderate', 'hard')
    lines, ret_dict = [], {}
    for j, curcls in enumerate(current_classes):
        name = _CLASS_TO_NAME[curcls]
        for i, kind in enumerate(('strict', 'loose')[:min_overlaps.shape[0]]):
            lines.append(_row(f'{name} AP@', min_overlaps[i, :, j], 2) + ':')
            lines += [_row(f'{m.label} AP:', m.values[j, :, i], m.digits) for m in table]
            # (dictionary order of the reference: per difficulty 3D, BEV, 2D)
            for idx, diff in enumerate(difficulty):
                for m in sorted((m for m in table if m.key), key=lambda m: ('3D', 'BEV', '2D').index(m.key)):
                    ret_dict[f'KITTI/{name}_{m.key}_{diff}_{kind}'] = m.values[j, idx, i]
    if len(current_classes) > 1:                                          # :756-779
        lines.append('\nOverall AP@' + ', '.join(difficulty) + ':')
        means = [_Metric(m.label, m.key, m.values.mean(axis=0), m.digits) for m in table]
        lines += [_row(f'{m.label} AP:', m.values[:, 0], m.digits) for m in means]
        for idx, diff in enumerate(difficulty):
            for m in sorted((m for m in means if m.key), key=lambda m: ('3D', 'BEV', '2D').index(m.key)):
                ret_dict[f'KITTI/Overall_{m.key}_{diff}'] = m.values[idx, 0]
    return ''.join(line + '\n' for line in lines), ret_dict
