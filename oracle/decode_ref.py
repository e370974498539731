"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the box decode reached through
``BEVHeight.get_bboxes`` (models/bev_height.py:116-126): mmdet3d 0.18.1 ``CenterHead.get_bboxes``,
``CenterPointBBoxCoder.decode`` (+ ``_topk``) and ``circle_nms``.

PARITY UNPINNED: mmdet3d is neither vendored in the reference nor installed here and the reference has
no fixture for this step; this follows the published algorithm.  Where the
upstream leaves the order of equal scores to torch.topk / numpy argsort, ties go to the lower flat
index (the device kernels do the same).
"""
import numpy as np


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32), dtype=np.float32))).astype(np.float32)


def _topk_desc(scores, k):
    """Indices of the k largest, descending, ties by lower index (stable)."""
    order = np.argsort(-scores, kind="stable")
    return order[:k]


def decode_task(pred, bbox_coder, test_cfg, task_id, norm_bbox=True):
    """pred: dict of numpy NCHW maps of ONE task -> list (per sample) of dict(bboxes, scores, labels)."""
    heat = _sigmoid(pred['heatmap'])
    B, cat, H, W = heat.shape
    K = bbox_coder['max_num']
    osf = np.float32(bbox_coder['out_size_factor'])
    vs = [np.float32(v) for v in bbox_coder['voxel_size']]
    pc = [np.float32(v) for v in bbox_coder['pc_range']]
    rng = bbox_coder.get('post_center_range')
    thr = bbox_coder.get('score_threshold')
    out = []
    for b in range(B):
        # _topk: per class top-K over H*W, then top-K over classes*K
        cs, ci = [], []
        for c in range(cat):
            flat = heat[b, c].reshape(-1)
            idx = _topk_desc(flat, K)
            cs.append(flat[idx])
            ci.append(idx)
        cs, ci = np.concatenate(cs), np.concatenate(ci)
        sel = _topk_desc(cs, K)
        scores = cs[sel]
        clses = (sel // K).astype(np.int64)
        inds = ci[sel]
        ys = (inds.astype(np.float32) / np.float32(W)).astype(np.int32).astype(np.float32)
        xs = (inds % W).astype(np.float32)
        g = lambda m, ch: m[b, ch].reshape(-1)[inds].astype(np.float32)
        xs = ((xs + g(pred['reg'], 0)) * osf * vs[0] + pc[0]).astype(np.float32)
        ys = ((ys + g(pred['reg'], 1)) * osf * vs[1] + pc[1]).astype(np.float32)
        hei = g(pred['height'], 0)
        dim = np.stack([g(pred['dim'], i) for i in range(3)], 1)
        if norm_bbox:
            dim = np.exp(dim, dtype=np.float32)
        rot = np.arctan2(g(pred['rot'], 0), g(pred['rot'], 1)).astype(np.float32)
        cols = [xs, ys, hei, dim[:, 0], dim[:, 1], dim[:, 2], rot]
        if 'vel' in pred:
            cols += [g(pred['vel'], 0), g(pred['vel'], 1)]
        boxes = np.stack(cols, 1).astype(np.float32)
        mask = np.ones(K, bool)
        if thr is not None:
            mask &= scores > np.float32(thr)
        if rng is not None:
            r = np.asarray(rng, np.float32)
            mask &= (boxes[:, :3] >= r[:3]).all(1) & (boxes[:, :3] <= r[3:]).all(1)
        boxes, scores, clses = boxes[mask], scores[mask], clses[mask]
        keep = circle_nms(np.concatenate([boxes[:, :2], scores[:, None]], 1), test_cfg['min_radius'][task_id],
                          test_cfg['post_max_size'])
        out.append(dict(bboxes=boxes[keep], scores=scores[keep], labels=clses[keep]))
    return out


def circle_nms(dets, thresh, post_max_size=83):
    x1, y1, scores = dets[:, 0], dets[:, 1], dets[:, 2]
    order = np.argsort(-scores, kind="stable")
    n = dets.shape[0]
    suppressed = np.zeros(n, np.int32)
    keep = []
    thresh = np.float32(thresh)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            dist = np.float32((x1[i] - x1[j]) ** 2 + (y1[i] - y1[j]) ** 2)
            if dist <= thresh:
                suppressed[j] = 1
    return np.asarray(keep[:post_max_size], np.int64)


def get_bboxes(preds, bbox_coder, test_cfg, num_classes, norm_bbox=True):
    """preds: tuple(task -> [dict(name -> numpy NCHW)]) -> list per sample of [boxes[n,9], scores[n], labels[n]]."""
    rets = [decode_task(p[0], bbox_coder, test_cfg, t, norm_bbox) for t, p in enumerate(preds)]
    B = len(rets[0])
    out = []
    for i in range(B):
        boxes = np.concatenate([r[i]['bboxes'] for r in rets])
        boxes = boxes.copy()
        boxes[:, 2] = boxes[:, 2] - boxes[:, 5] * np.float32(0.5)
        scores = np.concatenate([r[i]['scores'] for r in rets])
        flag, labels = 0, []
        for j, nc in enumerate(num_classes):
            labels.append(rets[j][i]['labels'] + flag)
            flag += nc
        out.append([boxes, scores, np.concatenate(labels).astype(np.int32)])
    return out
