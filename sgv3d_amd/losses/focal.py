"""``FocalLoss`` of losses/focal.py:12-90 on the MI355X: one HIP pass computes the loss value and the gradient with
respect to the logits (csrc/bsm_train.hip: ``sgv3d_focal_loss_with_logits``); the partial sums are float64 and added
in a fixed order, so a value is bitwise repeatable.

Constructor and ``forward(y_pred, y_true)`` keep the reference's meaning:

* ``mode='multiclass'`` (what exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:249 builds): ``y_pred`` logits
  ``[N, C, H, W]`` (any strides -- the channel-last maps of the training forward are read in place), ``y_true`` integer
  labels ``[N, H, W]``; for every class the binary focal loss of ``y_pred[:, c]`` against ``y_true == c`` with the
  chosen reduction, summed over the classes (focal.py:70-88); labels equal to ``ignore_index`` are left out;
* ``mode='binary' | 'multilabel'``: both tensors flattened (focal.py:59-68).

``normalized=True`` and ``reduced_threshold`` (losses/_functional.py:72-86) are not used by any shipped config and
raise ``NotImplementedError``; ``reduction`` is 'mean' or 'sum'.  CPU tensors raise: there is no CPU path.
"""
import torch
from torch.nn.modules.loss import _Loss

from .. import _lib
from ..hip_ops import prof

__all__ = ['FocalLoss', 'focal_loss_with_logits', 'BINARY_MODE', 'MULTICLASS_MODE', 'MULTILABEL_MODE']

# the three target layouts FocalLoss understands (the reference keeps them in losses/constants.py): one foreground
# channel / mutually exclusive class ids / independent per-channel targets
BINARY_MODE, MULTICLASS_MODE, MULTILABEL_MODE = "binary", "multiclass", "multilabel"


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, layout, kind, alpha, gamma, ignore_index, mean):
        batch, classes, pixels, sb, sc, sp = layout
        lib = _lib.load()
        dev = logits.device
        nws = lib.sgv3d_focal_loss_workspace_bytes()
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        grad = None
        if ctx.needs_input_grad[0]:
            # written in one pass together with the value, addressed like the logits (same storage geometry)
            grad = torch.empty_strided(logits.size(), logits.stride(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), prof("focal_loss"):
            rc = lib.sgv3d_focal_loss_with_logits(
                batch, classes, pixels, logits.data_ptr(), sb, sc, sp, target.data_ptr(), kind,
                -1.0 if alpha is None else float(alpha), float(gamma), 0 if ignore_index is None else int(ignore_index),
                0 if ignore_index is None else 1, 1 if mean else 0, 1.0, _lib.ptr(grad), out.data_ptr(), ws.data_ptr(), nws,
                _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_focal_loss_with_logits")
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None, None


def _require(t):
    if not t.is_cuda:
        raise RuntimeError("sgv3d_amd.losses runs on the MI355X only (no CPU fallback)")


def focal_loss_with_logits(output, target, gamma=2.0, alpha=0.25, reduction="mean", normalized=False,
                           reduced_threshold=None, eps=1e-6):
    """losses/_functional.py:37-108 for same-shape ``output`` / ``target`` (binary targets or soft targets)."""
    if normalized or reduced_threshold is not None:
        raise NotImplementedError("normalized / reduced focal loss (losses/_functional.py:72-86) is not built")
    if reduction not in ("mean", "sum"):
        raise NotImplementedError(f"reduction={reduction!r}: 'mean' and 'sum' are built")
    _require(output)
    x = output.float().contiguous().view(-1)
    t = target.to(device=x.device, dtype=torch.float32).contiguous().view(-1)
    assert x.numel() == t.numel() and x.numel() > 0
    return _Focal.apply(x, t, (1, 1, x.numel(), 0, 0, 1), 2, alpha, gamma, None, reduction == "mean")


class FocalLoss(_Loss):
    def __init__(self, mode, alpha=None, gamma=2.0, ignore_index=None, reduction="mean", normalized=False,
                 reduced_threshold=None):
        assert mode in {BINARY_MODE, MULTILABEL_MODE, MULTICLASS_MODE}
        super().__init__()
        if normalized or reduced_threshold is not None:
            raise NotImplementedError("normalized / reduced focal loss (losses/_functional.py:72-86) is not built")
        if reduction not in ("mean", "sum"):
            raise NotImplementedError(f"reduction={reduction!r}: 'mean' and 'sum' are built")
        self.mode = mode
        self.alpha = alpha
        self.gamma = gamma
        self.ignore_index = ignore_index
        self.reduction = reduction

    def forward(self, y_pred, y_true):
        _require(y_pred)
        if self.mode in {BINARY_MODE, MULTILABEL_MODE}:
            y_true = y_true.reshape(-1)
            y_pred = y_pred.reshape(-1)
            if self.ignore_index is not None:                     # focal.py:63-67
                keep = y_true != self.ignore_index
                y_pred, y_true = y_pred[keep], y_true[keep]
            return focal_loss_with_logits(y_pred, y_true, self.gamma, self.alpha, self.reduction)
        # multiclass: y_pred [N, C, *spatial] in any layout whose spatial dims are jointly contiguous
        assert y_pred.dim() >= 2 and y_pred.dtype == torch.float32
        N, C = int(y_pred.shape[0]), int(y_pred.shape[1])
        P = y_pred[0, 0].numel()
        assert tuple(y_true.shape) == (N,) + tuple(y_pred.shape[2:]), "y_true must be [N, *spatial] class ids"
        x = y_pred
        st = x.stride()
        if not all(st[i] == st[i + 1] * x.shape[i + 1] for i in range(2, x.dim() - 1)):
            x = x.contiguous()
            st = x.stride()
        sp = st[-1] if x.dim() > 2 else 1
        if y_true.dtype == torch.uint8:
            kind, lab = 0, y_true.contiguous()
        else:
            kind, lab = 1, y_true.long().contiguous()
        return _Focal.apply(x, lab, (N, C, P, st[0], st[1], sp), kind, self.alpha, self.gamma, self.ignore_index,
                            self.reduction == "mean")
