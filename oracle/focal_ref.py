"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by sgv3d_amd/).

CPU restatement (torch, float64 by default) of the semantic supervision of the SGV3D BSM experiment:

  losses/_functional.py:37-108   focal_loss_with_logits          -> focal_loss_with_logits
  losses/focal.py:57-90          FocalLoss.forward               -> focal_loss (binary / multilabel / multiclass)
  exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:258-276  get_downsampled_gt_semantic -> downsample_gt_semantic
  exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:291-302  get_loss                    -> semantic_loss

Pinned by tests/golden/losses.npz: values and input gradients of the reference's own ``losses.focal.FocalLoss``
executed in the build container (tests/golden/make_golden_aux.py).  The two experiment-file helpers cannot be imported
(the module needs pytorch_lightning / mmcv / nuscenes); they are a reshape + max and an interpolate + two loss calls.
"""
import torch
import torch.nn.functional as F


def focal_loss_with_logits(output, target, gamma=2.0, alpha=0.25, reduction="mean"):
    target = target.to(output.dtype)
    logpt = F.binary_cross_entropy_with_logits(output, target, reduction="none")      # :69
    pt = torch.exp(-logpt)                                                            # :70
    loss = (1.0 - pt).pow(gamma) * logpt                                              # :73-79
    if alpha is not None:
        loss = loss * (alpha * target + (1 - alpha) * (1 - target))                   # :81-82
    return loss.mean() if reduction == "mean" else loss.sum()                         # :88-91


def focal_loss(y_pred, y_true, mode="multiclass", alpha=None, gamma=2.0, ignore_index=None, reduction="mean"):
    if mode in ("binary", "multilabel"):                                              # focal.py:59-68
        y_true, y_pred = y_true.reshape(-1), y_pred.reshape(-1)
        if ignore_index is not None:
            keep = y_true != ignore_index
            y_pred, y_true = y_pred[keep], y_true[keep]
        return focal_loss_with_logits(y_pred, y_true, gamma, alpha, reduction)
    loss = 0                                                                          # focal.py:70-88
    keep = None if ignore_index is None else (y_true != ignore_index)
    for cls in range(y_pred.size(1)):
        t = (y_true == cls).long()
        p = y_pred[:, cls, ...]
        if keep is not None:
            t, p = t[keep], p[keep]
        loss = loss + focal_loss_with_logits(p, t, gamma, alpha, reduction)
    return loss


def downsample_gt_semantic(gt, downsample):
    """[B, N, H, W] class ids -> [B*N, H/d, W/d] int64: maximum id of each d x d block (:258-276)."""
    B, N, H, W = gt.shape
    g = gt.reshape(B * N, H // downsample, downsample, W // downsample, downsample)
    return g.permute(0, 1, 3, 2, 4).reshape(B * N, H // downsample, W // downsample, -1).max(-1).values.long()


def semantic_loss(img_preds, gt_semantic, downsample=8, alpha=0.25, gamma=2.0):
    """get_loss (:291-302): ((focal(upsample2x(semantic0)) + focal(semantic1)) / 2 against the block-max labels."""
    s0 = F.interpolate(img_preds[0], scale_factor=2, mode='bilinear')
    labels = downsample_gt_semantic(gt_semantic, downsample)
    return (focal_loss(s0, labels, "multiclass", alpha, gamma) + focal_loss(img_preds[1], labels, "multiclass", alpha, gamma)) / 2
