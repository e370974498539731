"""Training-mode BatchNorm2d fused with the residual add and ReLU around it, on NHWC float32 CUDA tensors
(csrc/bn_train.hip).  ``batch_norm_act`` is differentiable in the input, the residual and the
affine parameters and updates the module's running statistics like ``nn.BatchNorm2d`` in training mode."""
import os

import torch

from . import _lib, grad_slots
from .hip_ops import prof

__all__ = ['batch_norm_act', 'batch_norm_act_multi', 'deferred_counters']


MASK_FROM_X = os.environ.get("SGV3D_BN_MASK_FROM_X", "1") != "0"   # 0: the backward of relu(bn(x)) always reads the forward output


def _ws(channels, device):
    n = _lib.load().sgv3d_batchnorm_workspace_bytes(int(channels))
    return torch.empty(n, dtype=torch.uint8, device=device), n


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
        C = int(x.shape[-1])
        pixels = x.numel() // C
        if residual is not None:
            assert residual.shape == x.shape and residual.is_contiguous() and residual.dtype == torch.float32
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws, nws = _ws(C, x.device)
        with torch.cuda.device(x.device), prof("batchnorm_train_forward"):
            rc = _lib.load().sgv3d_batchnorm_train_forward(
                pixels, C, x.data_ptr(), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean),
                _lib.ptr(running_var), float(momentum), float(eps), 1 if relu else 0, y.data_ptr(), mean.data_ptr(),
                invstd.data_ptr(), ws.data_ptr(), nws, _lib.stream_handle(x.device))
        _lib.check(rc, "sgv3d_batchnorm_train_forward")
        # ReLU without a residual: the backward recomputes the mask from x (sgv3d_batchnorm_relu_train_backward_from_x), y is not kept
        ctx.from_x = bool(relu) and residual is None and MASK_FROM_X
        ctx.save_for_backward(x, y if (relu and not ctx.from_x) else None, weight, mean, invstd, bias)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        ctx.affine_versions = (None if weight is None else weight._version, None if bias is None else bias._version)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd, bias = ctx.saved_tensors
        dy = dy.contiguous()
        C = int(x.shape[-1])
        pixels = x.numel() // C
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[1]) else None
        # (the affine parameters' slots in the flat gradient buckets when they are free, grad_slots)
        dgamma = grad_slots.claim(weight) if ctx.needs_input_grad[2] else None
        dbeta = grad_slots.claim(bias) if ctx.needs_input_grad[3] else None
        if dgamma is None or dgamma.data_ptr() % 16:
            dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        if dbeta is None or dbeta.data_ptr() % 16:
            dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws, nws = _ws(C, x.device)
        if ctx.from_x:
            if ctx.affine_versions != (None if weight is None else weight._version, None if bias is None else bias._version):
                raise RuntimeError("batch_norm_act: the BatchNorm weight / bias were modified in place between forward and backward; the "
                                   "ReLU mask is recomputed from them (set SGV3D_BN_MASK_FROM_X=0 to keep the forward output instead)")
            with torch.cuda.device(x.device), prof("batchnorm_train_backward"):
                rc = _lib.load().sgv3d_batchnorm_relu_train_backward_from_x(
                    pixels, C, x.data_ptr(), dy.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), mean.data_ptr(), invstd.data_ptr(),
                    dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), nws, _lib.stream_handle(x.device))
            _lib.check(rc, "sgv3d_batchnorm_relu_train_backward_from_x")
            return (dx, None, dgamma if weight is not None else None, dbeta if ctx.needs_input_grad[3] else None,
                    None, None, None, None, None)
        with torch.cuda.device(x.device), prof("batchnorm_train_backward"):
            rc = _lib.load().sgv3d_batchnorm_train_backward(
                pixels, C, x.data_ptr(), _lib.ptr(y), dy.data_ptr(), _lib.ptr(weight), mean.data_ptr(), invstd.data_ptr(),
                1 if ctx.relu else 0, dx.data_ptr(), _lib.ptr(dres), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), nws,
                _lib.stream_handle(x.device))
        _lib.check(rc, "sgv3d_batchnorm_train_backward")
        return (dx, dres, dgamma if weight is not None else None, dbeta if ctx.needs_input_grad[3] else None,
                None, None, None, None, None)


class deferred_counters:
    """``with deferred_counters():`` -- the ``num_batches_tracked += 1`` of every batch_norm_act inside becomes ONE multi-tensor
    launch at the end of the block (126 one-element launches per forward of the R50 model otherwise)."""
    _open = None

    def __enter__(self):
        self._outer = deferred_counters._open
        deferred_counters._open = self.pending = []
        return self

    def __exit__(self, *exc):
        deferred_counters._open = self._outer
        if self.pending:
            # a module called twice in the block (shared BatchNorm, several sweeps) is in the list twice; the multi-tensor
            # kernel reads every operand before it writes any, so a repeated tensor would be incremented once.  One entry per
            # tensor, with its multiplicity as the per-tensor scalar.
            counts = {}
            for t in self.pending:
                entry = counts.setdefault(id(t), [t, 0])
                entry[1] += 1
            tensors = [e[0] for e in counts.values()]
            if all(e[1] == 1 for e in counts.values()):
                torch._foreach_add_(tensors, 1)
            else:
                torch._foreach_add_(tensors, [e[1] for e in counts.values()])
        return False


def _bump_version(t):
    """Mark ``t`` as modified in place (a kernel wrote it through its raw pointer) without launching anything.  (An in-place op on an
    empty view, the previous form, still launches a kernel -- 182 launches of ~5 us per training step of the R50 model.)"""
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:                              # (older torch)
        t[:0].add_(0)


def batch_norm_act(bn, x, residual=None, relu=False):
    """``relu(bn(x) + residual)`` for an ``nn.BatchNorm2d`` in training mode; ``x`` / ``residual`` NHWC float32."""
    assert bn.training and bn.track_running_stats, "training-mode BatchNorm with running statistics"
    if bn.num_batches_tracked is not None:
        if deferred_counters._open is not None and bn.momentum is not None:
            deferred_counters._open.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    if bn.momentum is None:
        # torch: cumulative moving average, factor 1 / num_batches_tracked (after the increment)
        if bn.num_batches_tracked is None:
            raise NotImplementedError("BatchNorm with momentum=None needs num_batches_tracked")
        momentum = 1.0 / float(bn.num_batches_tracked)          # one device->host read; momentum=None is not a configuration the reference uses
    else:
        momentum = bn.momentum
    out = _BatchNormAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, relu)
    # the kernel wrote the running statistics through raw pointers: bump their version counters so that anything keyed
    # on tensor versions (BEVHeight._stamp -> repacked inference weights) sees the change
    _bump_version(bn.running_mean)
    _bump_version(bn.running_var)
    return out


def batch_norm_act_multi(bns, x, relu=False):
    """``relu(bn_i(x[..., i * C:(i + 1) * C]))`` for n BatchNorm2d modules of one width C in training mode, on the n * C channels of ONE
    map: BatchNorm is per channel, so the n modules are one launch sequence over the wide map (their affine parameters concatenated
    through ``conv_grad.cat_params``, their running statistics gathered before and scattered back after the kernels).  The 36 hidden
    maps of the CenterHead branches: 6 launches instead of 216 per step."""
    from .conv_grad import cat_params
    n = len(bns)
    C = int(bns[0].num_features)
    assert int(x.shape[-1]) == n * C and all(b.training and b.track_running_stats and b.num_features == C and b.affine for b in bns)
    assert all(b.eps == bns[0].eps and b.momentum == bns[0].momentum and b.momentum is not None for b in bns), "one eps / momentum for all"
    for b in bns:
        if b.num_batches_tracked is not None:
            if deferred_counters._open is not None:
                deferred_counters._open.append(b.num_batches_tracked)
            else:
                b.num_batches_tracked.add_(1)
    gamma, beta = cat_params([b.weight for b in bns]), cat_params([b.bias for b in bns])
    with torch.no_grad():
        rm = torch.cat([b.running_mean for b in bns])
        rv = torch.cat([b.running_var for b in bns])
    out = _BatchNormAct.apply(x.contiguous(), None, gamma, beta, rm, rv, bns[0].momentum, bns[0].eps, relu)
    with torch.no_grad():
        torch._foreach_copy_([b.running_mean for b in bns], list(rm.split(C)))
        torch._foreach_copy_([b.running_var for b in bns], list(rv.split(C)))
    return out
