"""Differentiable forms of the two BSM-only elementwise layers (training side), NHWC float32 on
the MI355X: the x2 bilinear upsampling of TaskFPN (layers/backbones/bsm_lss_fpn.py:210) and the spatial-attention gate
with its residual, ``a + b * sigmoid(c)`` (bsm_lss_fpn.py:159,211).  Forward kernels: csrc/misc_layers.hip; adjoints:
csrc/bsm_train.hip."""
import torch

from . import _lib, hip_ops
from .hip_ops import prof

__all__ = ['upsample_bilinear2x', 'add_mul_sigmoid']


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hip_ops.upsample_bilinear2x(x)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, H2, W2, C = (int(v) for v in dy.shape)
        dx = torch.empty(B, H2 // 2, W2 // 2, C, dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device), prof("upsample_bilinear2x_backward"):
            rc = _lib.load().sgv3d_upsample_bilinear2x_backward(B, H2 // 2, W2 // 2, C, dy.data_ptr(), dx.data_ptr(),
                                                                _lib.stream_handle(dy.device))
        _lib.check(rc, "sgv3d_upsample_bilinear2x_backward")
        return dx


class _AddMulSigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(b, c)
        return hip_ops.add_mul_sigmoid(a, b, c)

    @staticmethod
    def backward(ctx, dy):
        b, c = ctx.saved_tensors
        dy = dy.contiguous()
        db, dc = torch.empty_like(b), torch.empty_like(c)
        with torch.cuda.device(dy.device), prof("add_mul_sigmoid_backward"):
            rc = _lib.load().sgv3d_add_mul_sigmoid_backward(dy.numel(), dy.data_ptr(), b.data_ptr(), c.data_ptr(),
                                                            db.data_ptr(), dc.data_ptr(), _lib.stream_handle(dy.device))
        _lib.check(rc, "sgv3d_add_mul_sigmoid_backward")
        return dy, db, dc


def upsample_bilinear2x(x):
    """``F.interpolate(scale_factor=2, mode='bilinear')`` of an NHWC map ``[B, H, W, C]`` (any strides are copied)."""
    return _Upsample2x.apply(x.contiguous())


def add_mul_sigmoid(a, b, c):
    """``a + b * sigmoid(c)`` for same-shape NHWC maps (element count a multiple of 4)."""
    return _AddMulSigmoid.apply(a.contiguous(), b.contiguous(), c.contiguous())
