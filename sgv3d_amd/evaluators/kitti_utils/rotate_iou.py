"""Rotated-box overlaps on the MI355X (csrc/rotate_iou.hip), behind the reference's entry point
``rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0)`` (evaluators/kitti_utils/rotate_iou.py:340-378)
and, for whole validation sets, ``rotate_iou_pairs``: every same-image pair in one launch instead of a dense matrix
per part of images."""
import numpy as np
import torch

from ... import _lib

__all__ = ['rotate_iou_gpu_eval', 'rotate_iou_pairs']

_TILE = 16


def rotate_iou_pairs(boxes_list, qboxes_list, criterion=-1, device='cuda'):
    """``boxes_list[m]`` [N_m, D], ``qboxes_list[m]`` [K_m, D] (D = 5: x, y, dx, dy, angle; D = 7: camera-frame 3-D
    boxes) -> list of float32 arrays [N_m, K_m].  One H2D copy, one launch, one D2H copy."""
    M = len(boxes_list)
    assert M == len(qboxes_list) and M > 0
    dim = int(boxes_list[0].shape[1]) if boxes_list[0].ndim == 2 else 5
    n = np.array([len(b) for b in boxes_list], np.int64)
    k = np.array([len(q) for q in qboxes_list], np.int64)
    tiles = -(-n // _TILE) * -(-k // _TILE)
    off = lambda v: np.concatenate([[0], np.cumsum(v)])
    box_off, qbox_off, tile_off, out_off = off(n), off(k), off(tiles), off(n * k)
    total = int(out_off[-1])
    if total == 0:
        return [np.zeros((int(a), int(b)), np.float32) for a, b in zip(n, k)]
    dev = torch.device(device)
    cat = lambda lst: np.concatenate([np.asarray(x, np.float64).reshape(-1, dim) for x in lst], 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)
    d_boxes, d_q = t(cat(boxes_list), np.float64), t(cat(qboxes_list), np.float64)
    d_bo, d_qo, d_to, d_oo = t(box_off, np.int32), t(qbox_off, np.int32), t(tile_off, np.int32), t(out_off[:-1], np.int64)
    out = torch.empty(total, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sgv3d_rotate_iou_pairs(M, int(tile_off[-1]), d_bo.data_ptr(), d_qo.data_ptr(), d_to.data_ptr(),
                                                d_oo.data_ptr(), d_boxes.data_ptr(), d_q.data_ptr(), dim, int(criterion),
                                                out.data_ptr(), _lib.stream_handle(dev))
    _lib.check(rc, "sgv3d_rotate_iou_pairs")
    flat = out.cpu().numpy()
    return [flat[out_off[m]:out_off[m + 1]].reshape(int(n[m]), int(k[m])) for m in range(M)]


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """Dense [N, K] overlap matrix of BEV rectangles ``(x, y, dx, dy, angle)``; returned in ``boxes.dtype`` like the
    reference's (computed in float32)."""
    boxes = np.asarray(boxes)
    N, K = boxes.shape[0], np.asarray(query_boxes).shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float32)
    return rotate_iou_pairs([boxes], [np.asarray(query_boxes)], criterion, f'cuda:{device_id}')[0].astype(boxes.dtype)
