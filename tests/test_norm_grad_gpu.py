"""Fused training-mode BatchNorm (+ residual, + ReLU) kernels on the MI355X against torch's batch_norm in float64 on the
CPU: output, running statistics, and the gradients of input, residual, weight and bias."""
import pytest
import torch
import torch.nn.functional as F

from sgv3d_amd.norm_grad import batch_norm_act

pytestmark = pytest.mark.gpu


def _reference(x, res, bn, relu, dy):
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w = bn.weight.detach().double().cpu().requires_grad_(True)
    b = bn.bias.detach().double().cpu().requires_grad_(True)
    rm, rv = bn.running_mean.detach().double().cpu().clone(), bn.running_var.detach().double().cpu().clone()
    y = F.batch_norm(xr, rm, rv, w, b, True, bn.momentum, bn.eps)
    rr = None
    if res is not None:
        rr = res.double().permute(0, 3, 1, 2).requires_grad_(True)
        y = y + rr
    pre = y.detach()
    if relu:
        y = F.relu(y)
    y.backward(dy.double().permute(0, 3, 1, 2))
    p = lambda t: t.permute(0, 2, 3, 1)
    return p(y.detach()), p(xr.grad), (p(rr.grad) if rr is not None else None), w.grad, b.grad, rm, rv, p(pre)


@pytest.mark.parametrize("shape,use_res,relu,offset", [
    ((2, 20, 28, 64), False, True, 0.0),
    ((1, 13, 17, 256), True, True, 0.0),
    ((3, 9, 11, 128), True, False, 0.0),
    ((2, 31, 33, 80), False, False, 0.0),
    ((2, 40, 48, 64), False, True, 25.0),        # |mean| >> std: E[x^2] - mean^2 must not cancel
    ((2, 1, 1, 512), False, True, 0.0),          # the ASPP's pooled branch: two "pixels"
    # full-size maps: 1024 partial rows per channel; 102 rows (ragged); four channel groups of 256 rows
    ((2, 216, 384, 64), False, True, 0.0),
    ((1, 100, 131, 64), True, True, 0.0),
    ((1, 256, 260, 256), False, True, 3.0),
])
def test_batch_norm_act_matches_torch(shape, use_res, relu, offset):
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[-1]
    x = torch.randn(shape, generator=g) * 0.7 + offset
    res = torch.randn(shape, generator=g) if use_res else None
    dy = torch.randn(shape, generator=g)
    bn = torch.nn.BatchNorm2d(C, momentum=0.1).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    want = _reference(x, res, bn, relu, dy)
    xg = x.cuda().requires_grad_(True)
    rg = res.cuda().requires_grad_(True) if use_res else None
    y = batch_norm_act(bn, xg, rg, relu)
    y.backward(dy.cuda())
    tol = lambda w: 3e-5 * max(1.0, float(w.abs().max()))
    assert float((y.detach().cpu().double() - want[0]).abs().max()) <= tol(want[0])
    # an element whose pre-activation is within f32 rounding of zero may take the other side of the ReLU than the float64 reference
    # (17 M elements: one or two do); its gradient is then dy instead of 0 -- not an error of the kernels.  At most a handful.
    sure = (want[7].abs() > 1e-5) if relu else torch.ones_like(want[7], dtype=torch.bool)
    assert int((~sure).sum()) <= max(8, sure.numel() // 10 ** 5)
    assert float(((xg.grad.cpu().double() - want[1]).abs() * sure).max()) <= tol(want[1])
    if use_res:
        assert float(((rg.grad.cpu().double() - want[2]).abs() * sure).max()) <= tol(want[2])
    assert float((bn.weight.grad.cpu().double() - want[3]).abs().max()) <= 1e-4 * max(1.0, float(want[3].abs().max()))
    assert float((bn.bias.grad.cpu().double() - want[4]).abs().max()) <= 1e-4 * max(1.0, float(want[4].abs().max()))
    assert float((bn.running_mean.cpu().double() - want[5]).abs().max()) <= 1e-5 * max(1.0, abs(offset))
    assert float((bn.running_var.cpu().double() - want[6]).abs().max()) <= 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_batch_norm_act_is_deterministic():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 37, 41, 192, generator=g).cuda()
    dy = torch.randn(2, 37, 41, 192, generator=g).cuda()
    outs = []
    for _ in range(2):
        bn = torch.nn.BatchNorm2d(192).cuda().train()
        xg = x.clone().requires_grad_(True)
        y = batch_norm_act(bn, xg, None, True)
        y.backward(dy)
        outs.append((y.detach().clone(), xg.grad.clone(), bn.weight.grad.clone(), bn.running_var.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("shape", [(2, 37, 41, 192), (1, 64, 80, 64), (2, 1, 1, 512)])
def test_relu_mask_recomputed_from_x_is_the_forward_mask(shape):
    """relu(bn(x)) without a residual: the backward derives the mask from x (sgv3d_batchnorm_relu_train_backward_from_x, the
    forward's scale / shift with the forward's roundings) instead of reading y -- every gradient bitwise what the y-mask path
    gives, including elements that sit exactly on the threshold (inputs on a coarse grid)."""
    from sgv3d_amd import norm_grad
    g = torch.Generator().manual_seed(shape[-1])
    x = (torch.randn(shape, generator=g) * 4).round() / 4                  # many equal values, some of them mapping to y == 0 exactly
    dy = torch.randn(shape, generator=g).cuda()
    outs = []
    for from_x in (True, False):
        old = norm_grad.MASK_FROM_X
        norm_grad.MASK_FROM_X = from_x
        try:
            bn = torch.nn.BatchNorm2d(shape[-1]).cuda().train()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, shape[-1]))
                bn.bias.copy_(torch.linspace(-0.4, 0.4, shape[-1]))
            xg = x.cuda().requires_grad_(True)
            y = batch_norm_act(bn, xg, None, True)
            y.backward(dy)
            outs.append((y.detach().clone(), xg.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
        finally:
            norm_grad.MASK_FROM_X = old
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_deferred_step_counters_and_version_bumps():
    """``deferred_counters``: the num_batches_tracked increments of every batch_norm_act inside the block happen once, at its end, in
    one multi-tensor launch; the running statistics the kernel wrote through raw pointers still count as modified (version bump
    without a launch)."""
    from sgv3d_amd.norm_grad import deferred_counters
    bns = [torch.nn.BatchNorm2d(32).cuda().train() for _ in range(3)]
    x = torch.randn(2, 5, 7, 32, device="cuda")
    versions = [(b.running_mean._version, b.running_var._version) for b in bns]
    with deferred_counters():
        for b in bns:
            batch_norm_act(b, x, None, True)
        batch_norm_act(bns[0], x, None, False)                              # the same module twice: counted twice
        assert all(int(b.num_batches_tracked) == 0 for b in bns)            # nothing yet
    assert [int(b.num_batches_tracked) for b in bns] == [2, 1, 1]
    for b, (vm, vv) in zip(bns, versions):
        assert b.running_mean._version > vm and b.running_var._version > vv
        assert float(b.running_mean.abs().max()) > 0
    batch_norm_act(bns[1], x, None, True)                                   # outside a block: immediately
    assert int(bns[1].num_batches_tracked) == 2

