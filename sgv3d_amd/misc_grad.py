"""Differentiable forms of the small layers of the training forward that ran as ATen operators in round 1: the stem's
max pooling, the ASPP image-pooling branch (global average + 1x1 convolution on a [B, C] vector) and the deformable
bilinear sampling of the DCN.  NHWC float32 on the MI355X; forward kernels in
csrc/misc_layers.hip, adjoints in csrc/train_misc.hip."""
import torch

from . import _lib, hip_ops
from .hip_ops import prof

__all__ = ['maxpool3x3s2', 'pooled_linear', 'deform_im2col3x3']


def _st(t):
    return _lib.stream_handle(t.device)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, C = (int(v) for v in x.shape)
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, oh, ow, C, dtype=torch.float32, device=x.device)
        idx = torch.empty(B, oh, ow, C, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device), prof("maxpool3x3s2_train"):
            rc = _lib.load().sgv3d_maxpool3x3s2_train_forward(B, H, W, C, x.data_ptr(), y.data_ptr(), idx.data_ptr(), _st(x))
        _lib.check(rc, "sgv3d_maxpool3x3s2_train_forward")
        ctx.save_for_backward(idx)
        ctx.in_shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, H, W, C = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device), prof("maxpool3x3s2_backward"):
            rc = _lib.load().sgv3d_maxpool3x3s2_backward(B, H, W, C, idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), _st(dy))
        _lib.check(rc, "sgv3d_maxpool3x3s2_backward")
        return dx


def maxpool3x3s2(x):
    """``nn.MaxPool2d(3, 2, 1)`` on an NHWC map (channels % 4 == 0)."""
    return _MaxPool.apply(x.contiguous())


class _PooledLinear(torch.autograd.Function):
    """y[b] = W @ mean_pixels(x[b]): AdaptiveAvgPool2d((1, 1)) + bias-free 1x1 Conv2d of the ASPP (lss_fpn.py:80-88)."""

    @staticmethod
    def forward(ctx, x, weight):
        B, H, W, C = (int(v) for v in x.shape)
        w2 = weight.reshape(weight.shape[0], -1).contiguous()
        pooled = hip_ops.global_avgpool(x)                                   # [B, C]
        y = hip_ops.dense(pooled, w2)                                        # [B, N]
        ctx.save_for_backward(pooled, w2)
        ctx.x_shape, ctx.w_shape = (B, H, W, C), tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        pooled, w2 = ctx.saved_tensors
        B, H, W, C = ctx.x_shape
        dy = dy.contiguous()
        N = int(w2.shape[0])
        dpooled = hip_ops.dense(dy, w2.t().contiguous())                     # [B, C]
        dw = torch.empty(N, C, dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device), prof("dense_backward_weight"):
            rc = _lib.load().sgv3d_dense_backward_weight(B, C, N, pooled.data_ptr(), dy.data_ptr(), dw.data_ptr(), _st(dy))
        _lib.check(rc, "sgv3d_dense_backward_weight")
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        hip_ops.broadcast_channels((dpooled * (1.0 / (H * W))).contiguous(), dx)   # every pixel receives d pooled / (H W)
        return dx, dw.reshape(ctx.w_shape)


def pooled_linear(x, weight):
    """x NHWC [B, H, W, C], weight [N, C, 1, 1] -> [B, N] = weight @ mean over pixels."""
    return _PooledLinear.apply(x.contiguous(), weight)


class _DeformIm2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, groups):
        col = hip_ops.deform_im2col3x3(x, offset, groups)
        ctx.save_for_backward(x, offset)
        ctx.groups = groups
        return col

    @staticmethod
    def backward(ctx, dcol):
        x, offset = ctx.saved_tensors
        B, H, W, C = (int(v) for v in x.shape)
        dcol = dcol.contiguous()
        dx = torch.empty_like(x)
        doff = torch.zeros_like(offset)
        with torch.cuda.device(x.device), prof("deform_im2col3x3_backward"):
            rc = _lib.load().sgv3d_deform_im2col3x3_backward(B, H, W, C, int(ctx.groups), x.data_ptr(), offset.data_ptr(),
                                                             int(offset.shape[-1]), dcol.data_ptr(), dx.data_ptr(),
                                                             doff.data_ptr(), int(doff.shape[-1]), _st(x))
        _lib.check(rc, "sgv3d_deform_im2col3x3_backward")
        return dx, doff, None


def deform_im2col3x3(x, offset, groups):
    """Deformable bilinear im2col of a 3x3 / pad 1 DCNv1: x NHWC [B,H,W,C], offset NHWC [B,H,W,>=18] ->
    col [B,H,W,groups*9*(C/groups)], differentiable in x and offset."""
    return _DeformIm2col.apply(x.contiguous(), offset.contiguous(), groups)
