"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the training-side head functions
(SURVEY.md §8(f) rank 2): target assignment and the detection loss.

* ``get_targets_single`` follows layers/heads/bev_height_head.py:113-253 statement by statement (the per-task
  regrouping of the boxes, the fp32 cell arithmetic, the int truncation of the centre, the slot = position in
  the regrouped list, the skipped-but-counted out-of-range boxes).
* ``gaussian_radius`` / ``draw_heatmap_gaussian`` / ``gaussian_2d`` are mmdet3d 0.18.1
  ``mmdet3d/core/utils/gaussian.py`` and ``get_targets`` is ``CenterHead.get_targets`` (multi_apply + stack).
* ``loss`` follows layers/heads/bev_height_head.py:255-311 with mmdet 2.19.0's ``GaussianFocalLoss``
  (alpha 2, gamma 4, eps 1e-12), ``L1Loss`` (loss_weight 0.25 in the experiment files), mmdet3d's
  ``clip_sigmoid`` (clamp to [1e-4, 1 - 1e-4]) and ``reduce_mean`` (identity on one rank).

PARITY UNPINNED for the mmdet / mmdet3d pieces: those wheels are neither vendored in the reference nor installed
here, and the reference has no tests or fixtures for these functions; they are restated from the published
definitions.
"""
import numpy as np

F = np.float32


def gaussian_radius(det_size, min_overlap=0.5):
    """fp32 like the 0-dim CUDA tensors the reference passes in (python scalars are cast to fp32 per op)."""
    height, width = F(det_size[0]), F(det_size[1])
    a1 = F(1)
    b1 = F(height + width)
    c1 = F(F(F(width * height) * F(1 - min_overlap)) / F(1 + min_overlap))
    sq1 = np.sqrt(F(F(b1 * b1) - F(F(4) * a1 * c1)), dtype=F)
    r1 = F(F(b1 + sq1) / F(2))
    a2 = F(4)
    b2 = F(F(2) * F(height + width))
    c2 = F(F(F(1 - min_overlap) * width) * height)
    sq2 = np.sqrt(F(F(b2 * b2) - F(F(4) * a2 * c2)), dtype=F)
    r2 = F(F(b2 + sq2) / F(2))
    a3 = F(4 * min_overlap)
    b3 = F(F(-2 * min_overlap) * F(height + width))
    c3 = F(F(F(min_overlap - 1) * width) * height)
    sq3 = np.sqrt(F(F(b3 * b3) - F(F(F(4) * a3) * c3)), dtype=F)
    r3 = F(F(b3 + sq3) / F(2))
    return min(r1, r2, r3)


def gaussian_2d(shape, sigma=1.0):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    diameter = 2 * radius + 1
    gaussian = gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right].astype(F)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        np.maximum(masked_heatmap, masked_gaussian * F(k), out=masked_heatmap)
    return heatmap


def get_targets_single(gt_boxes, gt_labels, class_names, train_cfg, norm_bbox=True):
    """gt_boxes f32 [N, 9] (x, y, z, w, l, h, yaw, vx, vy), gt_labels int [N]; ``class_names`` = list (per task)
    of lists of names.  Returns per-task lists (heatmaps, anno_boxes, inds, masks)."""
    gt_boxes = np.asarray(gt_boxes, F).reshape(-1, 9)
    gt_labels = np.asarray(gt_labels, np.int64).reshape(-1)
    max_objs = train_cfg['max_objs'] * train_cfg['dense_reg']
    grid_size = np.asarray(train_cfg['grid_size'])
    pc_range = np.asarray(train_cfg['point_cloud_range'], F)
    voxel_size = np.asarray(train_cfg['voxel_size'], F)
    osf = train_cfg['out_size_factor']
    fms = grid_size[:2] // osf
    task_boxes, task_classes = [], []
    flag = 0
    for names in class_names:
        idx = [np.where(gt_labels == i + flag)[0] for i in range(len(names))]
        task_boxes.append(np.concatenate([gt_boxes[m] for m in idx], 0))
        task_classes.append(np.concatenate([gt_labels[m] + 1 - flag for m in idx]).astype(np.int64))
        flag += len(names)
    heatmaps, anno_boxes, inds, masks = [], [], [], []
    for t, names in enumerate(class_names):
        heatmap = np.zeros((len(names), int(fms[1]), int(fms[0])), F)
        anno_box = np.zeros((max_objs, 10), F)
        ind = np.zeros(max_objs, np.int64)
        mask = np.zeros(max_objs, np.uint8)
        num_objs = min(task_boxes[t].shape[0], max_objs)
        for k in range(num_objs):
            box = task_boxes[t][k]
            cls_id = int(task_classes[t][k]) - 1
            width = F(F(box[3] / voxel_size[0]) / F(osf))
            length = F(F(box[4] / voxel_size[1]) / F(osf))
            if width > 0 and length > 0:
                radius = gaussian_radius((length, width), min_overlap=train_cfg['gaussian_overlap'])
                radius = max(train_cfg['min_radius'], int(radius))
                x, y, z = box[0], box[1], box[2]
                coor_x = F(F(F(x - pc_range[0]) / voxel_size[0]) / F(osf))
                coor_y = F(F(F(y - pc_range[1]) / voxel_size[1]) / F(osf))
                cx, cy = int(coor_x), int(coor_y)              # .to(torch.int32): truncation toward zero
                if not (0 <= cx < fms[0] and 0 <= cy < fms[1]):
                    continue
                draw_heatmap_gaussian(heatmap[cls_id], (cx, cy), radius)
                ind[k] = cy * int(fms[0]) + cx
                mask[k] = 1
                dim = box[3:6]
                if norm_bbox:
                    dim = np.log(dim, dtype=F)
                anno_box[k] = np.array([F(coor_x - F(cx)), F(coor_y - F(cy)), z, dim[0], dim[1], dim[2],
                                        np.sin(box[6], dtype=F), np.cos(box[6], dtype=F), box[7], box[8]], F)
        heatmaps.append(heatmap)
        anno_boxes.append(anno_box)
        inds.append(ind)
        masks.append(mask)
    return heatmaps, anno_boxes, inds, masks


def get_targets(gt_boxes_list, gt_labels_list, class_names, train_cfg, norm_bbox=True):
    """CenterHead.get_targets: per-sample targets stacked per task -> lists of [B, ...] arrays."""
    per = [get_targets_single(b, l, class_names, train_cfg, norm_bbox) for b, l in zip(gt_boxes_list, gt_labels_list)]
    out = []
    for field in range(4):
        out.append([np.stack([per[b][field][t] for b in range(len(per))]) for t in range(len(class_names))])
    return tuple(out)


def clip_sigmoid(x, eps=1e-4):
    x = np.asarray(x, np.float64)
    return np.clip(1.0 / (1.0 + np.exp(-x)), eps, 1 - eps)


def gaussian_focal_loss(pred, target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos = (target == 1).astype(np.float64)
    neg = (1 - target) ** gamma
    return -np.log(pred + eps) * (1 - pred) ** alpha * pos - np.log(1 - pred + eps) * pred ** alpha * neg


def loss(targets, preds, code_weights, loss_bbox_weight=0.25, world_mean=lambda v: v):
    """``preds``: per task a dict of numpy NCHW maps (heatmap = raw logits).  float64 arithmetic (the checker
    should be more precise than both implementations).  Returns (total, per-task [(heatmap, bbox)])."""
    heatmaps, anno_boxes, inds, masks = targets
    total, parts = 0.0, []
    cw = np.asarray(code_weights, np.float64)
    for t, p in enumerate(preds):
        heat = clip_sigmoid(p['heatmap'])
        tgt = heatmaps[t].astype(np.float64)
        num_pos = float((heatmaps[t] == 1).sum())
        avg = max(world_mean(num_pos), 1.0)
        l_heat = gaussian_focal_loss(heat, tgt).sum() / avg
        anno = np.concatenate([p['reg'], p['height'], p['dim'], p['rot'], p['vel']], 1).astype(np.float64)
        B, C, H, W = anno.shape
        flat = anno.transpose(0, 2, 3, 1).reshape(B, H * W, C)
        pred = np.take_along_axis(flat, inds[t][:, :, None].astype(np.int64), 1)
        m = masks[t].astype(np.float64)[:, :, None] * np.ones_like(anno_boxes[t], np.float64)
        m = m * (~np.isnan(anno_boxes[t])).astype(np.float64)
        num = max(world_mean(float(masks[t].astype(np.float64).sum())), 1e-4)
        l_box = (np.abs(pred - anno_boxes[t].astype(np.float64)) * (m * cw)).sum() / num * loss_bbox_weight
        total += l_box + l_heat
        parts.append((l_heat, l_box))
    return total, parts
