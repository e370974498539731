"""Semantic (SAM-mask) supervision of the SGV3D BSM experiment on the MI355X.

Follows ``get_downsampled_gt_semantic`` / ``get_semantic_loss`` / ``get_loss`` of
exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:258-305: the mask image ``gt_semantic`` (uint8 class ids,
``[B, num_cams, H, W]``, dataset/nusc_mv_det_dataset.py:603-615) is reduced to the stride-8 feature grid by the maximum
id of each 8x8 block; the stride-16 logits are bilinearly upsampled x2; both maps go through the multiclass focal loss
(alpha 0.25, gamma 2, mean) and the two values are averaged.  Three kernels instead of the reference's ~40 torch
kernels per map: ``sgv3d_semantic_labels_downsample``, ``sgv3d_upsample_bilinear2x`` (+ its adjoint) and
``sgv3d_focal_loss_with_logits``.
"""
import torch
from torch import nn

from .. import _lib
from ..bsm_grad import upsample_bilinear2x
from ..hip_ops import prof
from .focal import FocalLoss

__all__ = ['SemanticSupervision', 'downsample_gt_semantic']


def downsample_gt_semantic(gt_semantics, downsample):
    """uint8 ``[B, N, H, W]`` -> uint8 ``[B*N, H/downsample, W/downsample]`` (:258-276; the reference returns int64)."""
    if not gt_semantics.is_cuda:
        raise RuntimeError("sgv3d_amd.losses runs on the MI355X only (no CPU fallback)")
    B, N, H, W = (int(v) for v in gt_semantics.shape)
    g = gt_semantics.to(torch.uint8).contiguous()
    out = torch.empty(B * N, H // downsample, W // downsample, dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device), prof("semantic_labels_downsample"):
        rc = _lib.load().sgv3d_semantic_labels_downsample(B * N, H, W, int(downsample), g.data_ptr(), out.data_ptr(),
                                                          _lib.stream_handle(g.device))
    _lib.check(rc, "sgv3d_semantic_labels_downsample")
    return out


class SemanticSupervision(nn.Module):
    """``get_loss(img_preds, gt_semantic)`` of the BSM experiment (:291-302).  ``img_preds`` = (semantic0, semantic1):
    the logits ``[B*N, classes, h, w]`` the training forward returns (NCHW views of channel-last buffers)."""

    def __init__(self, downsample=8, alpha=0.25, gamma=2.0):
        super().__init__()
        self.downsample = downsample
        self.focal_loss = FocalLoss(mode='multiclass', alpha=alpha, gamma=gamma, reduction="mean")    # :249

    def forward(self, img_preds, gt_semantic):
        semantic0, semantic1 = img_preds
        up = upsample_bilinear2x(semantic0.permute(0, 2, 3, 1))             # NHWC in, NHWC out (:293)
        labels = downsample_gt_semantic(gt_semantic, self.downsample)       # :295
        loss0 = self.focal_loss(up.permute(0, 3, 1, 2), labels)             # :297
        loss1 = self.focal_loss(semantic1, labels)                          # :298
        return (loss0 + loss1) / 2                                          # :299
