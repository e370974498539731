"""Training-mode forward / backward of the whole detector on the MI355X against the oracle's training-mode
restatement in float64 on the CPU: predictions, loss and the gradient of every parameter."""
import numpy as np
import pytest
import torch

from oracle import torch_model as O
from sgv3d_amd import synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.train_step import DataParallelAdamW

pytestmark = pytest.mark.gpu


def _model(seed=0):
    torch.manual_seed(seed)
    bconf, hconf = synthetic.small_conf()
    model = BEVHeight(bconf, hconf)
    synthetic.randomize_norm_stats_(model, seed=seed)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                                           # the ASPP's Dropout(0.5) would need a shared RNG stream
    return model, bconf, hconf


def _gt(B, bev_cells=64):
    boxes, labels = synthetic.make_gt(B, seed=1, n_range=(6, 12), stress=False)
    for b in boxes:                                             # the reduced grid covers 25.6 m x 25.6 m
        b[:, 0] = b[:, 0] * 0.25
        b[:, 1] = b[:, 1] * 0.25
    return boxes, labels


def _oracle_loss(preds, targets, code_weights):
    total = 0
    for t, pl in enumerate(preds):
        p = pl[0]
        heat = torch.clamp(torch.sigmoid(p['heatmap']), 1e-4, 1 - 1e-4)
        tgt = targets[0][t]
        pos = tgt.eq(1).double()
        l = -(heat + 1e-12).log() * (1 - heat) ** 2 * pos - (1 - heat + 1e-12).log() * heat ** 2 * (1 - tgt) ** 4
        total = total + l.sum() / max(float(pos.sum()), 1.0)
        anno = torch.cat([p['reg'], p['height'], p['dim'], p['rot'], p['vel']], 1)
        B, C, H, W = anno.shape
        pred = anno.permute(0, 2, 3, 1).reshape(B, H * W, C).gather(1, targets[2][t][:, :, None].expand(-1, -1, C))
        m = targets[3][t].double()[:, :, None] * torch.tensor(code_weights, dtype=torch.float64)
        total = total + ((pred - targets[1][t]).abs() * m).sum() / max(float(targets[3][t].sum()), 1e-4) * 0.25
    return total


def _compare_with_oracle(spread_regressions=False):
    """One training-mode forward / backward of the small model on the GPU and in float64 through the oracle: (worst relative
    prediction error over the maps, relative loss error, [(relative L2 gradient error, name, |want|)] sorted descending).
    ``spread_regressions``: the final layers of the regression branches get N(0, 1) biases, so that the L1 loss's residuals are
    O(1) instead of the ~0 of untrained branches against zero targets (see the bf16 test)."""
    B = 2
    model, bconf, hconf = _model()
    if spread_regressions:
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for t in model.head.task_heads:
                for k in ('reg', 'height', 'dim', 'rot', 'vel'):
                    getattr(t, k)[-1].bias.copy_(torch.randn(getattr(t, k)[-1].bias.shape, generator=g))
    model = model.cuda().train()
    head_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
    model.head.train_cfg = head_cfg
    imgs = synthetic.make_images(B, final=bconf['final_dim'], device='cuda', seed=3)
    mats = synthetic.make_mats(B, device='cuda')
    boxes, labels = _gt(B)
    preds = model(imgs, mats)
    targets = model.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    loss = model.loss(targets, preds)
    loss.backward()
    grads = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None}

    sd = {k: (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu()) for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    for n in names:
        sd[n].requires_grad_(True)
    rpreds = O.bevheight_train_forward(sd, bconf, hconf, imgs.cpu(), {k: v.cpu() for k, v in mats.items()})
    tc = tuple([x.cpu().double() if x.dtype == torch.float32 else x.cpu() for x in part] for part in targets)
    rloss = _oracle_loss(rpreds, tc, head_cfg['code_weights'])
    rloss.backward()

    pred_err = 0.0
    for t, (pl, rl) in enumerate(zip(preds, rpreds)):
        for k in pl[0]:
            got, want = pl[0][k].detach().cpu().double(), rl[0][k].detach()
            pred_err = max(pred_err, float((got - want).abs().max()) / max(1.0, float(want.abs().max())))
    loss_err = abs(float(loss.detach()) - float(rloss.detach())) / abs(float(rloss.detach()))
    worst = []
    for n in names:
        want = sd[n].grad
        if want is None:
            assert n not in grads or float(grads[n].abs().max()) == 0, n      # assist_layer: unused on this path
            continue
        got = grads[n]
        rel = float((got - want).norm() / (want.norm() + 1e-12))
        worst.append((rel, n, float(want.norm())))
    worst.sort(reverse=True)
    return pred_err, loss_err, worst


def test_training_forward_backward_matches_oracle():
    pred_err, loss_err, worst = _compare_with_oracle()
    assert pred_err <= 2e-3 and loss_err <= 1e-3, (pred_err, loss_err)
    print('largest relative gradient errors:', [(f'{r:.1e}', n) for r, n, _ in worst[:5]])
    bad = [w for w in worst if w[0] > 5e-3 and w[2] > 1e-7]
    assert not bad, bad[:10]
    assert sorted(w[0] for w in worst)[len(worst) // 2] < 3e-4          # typical tensor: fp32 rounding only
    assert len(worst) > 150


# Mixed-precision step (tools/train_bench.py --dtype bf16; BASELINE configs[4] names bf16, the reference trains with
# --amp_backend native, docs/run_and_eval.md:5,16): every convolution product -- forward, data gradient, weight gradient -- on the
# bf16 matrix cores with f32 accumulation; parameters (the master copies), activations, BatchNorm statistics, loss, gradients and
# AdamW state stay f32.  Stated tolerance against the float64 oracle, per gradient TENSOR (relative L2 error): each operand
# rounding is 2^-9 relative and averages out over a layer's sum, so a typical tensor sits at a few 1e-3; the tensors behind
# training-mode BatchNorms on the smallest maps (the BEV trunk's last stages: 8 x 8 cells x 2 samples per channel at this test's
# size) are the tail -- the BatchNorm adjoint subtracts the batch means of dy and dy x-hat, which leaves the rounding noise of the
# whole tensor on a small difference.  The backward of this untrained batch-2 network is ill-conditioned at ANY precision: the
# f32 kernels' own errors (~1e-6 per operation) arrive at 3e-4 typical / 1.2e-3 worst per tensor in the test above, an
# amplification of ~300 that the 2^-9 roundings of bf16 see as well.  Measured here (printed by the test): predictions 2.6e-2
# of their scale, loss 2.4e-4, gradient tensors median 3.0e-2, 90th percentile 0.21, worst 0.26 (cosine to the oracle 0.97).
# Bars, ~2x head-room: predictions 5e-2, loss 2e-3, median 6e-2, 90th percentile 0.35, every tensor 0.5; and, the functional
# statement, test_bf16_training_steps_track_the_f32_steps: six AdamW steps land within 2 % of the f32 run's loss.
BF16_TRAIN_TOL = dict(pred=5e-2, loss=2e-3, grad_median=6e-2, grad_p90=0.35, grad_max=0.5)


def test_training_forward_backward_bf16_mode():
    from sgv3d_amd import hip_ops
    saved = hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
    try:
        hip_ops.PROFILE = []
        pred_err, loss_err, worst = _compare_with_oracle(spread_regressions=True)
        kernels = {r[0].split('|')[0] for r in hip_ops.PROFILE}
    finally:
        hip_ops.PROFILE = None
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = saved
    assert "conv_wgrad_bf16" in kernels and any(k.startswith("conv_igemm_bf16") or k.startswith("conv_patch") for k in kernels), sorted(kernels)
    errs = sorted(w[0] for w in worst if w[2] > 1e-7)
    med, p90 = errs[len(errs) // 2], errs[int(0.9 * len(errs))]
    # (The regression branches are given O(1) residuals: with untrained branches against zero targets the L1 loss's gradient is
    #  sign(prediction) of numbers smaller than one bf16 rounding -- two correct executions at different precisions disagree on
    #  those signs, and the disagreement, not the arithmetic, was what a first version of this test measured: 0.3-0.6 on the
    #  velocity / size branches and on the trunk layers behind them.)
    sig = [w for w in worst if w[2] > 1e-7]
    print(f"bf16 training mode vs float64 oracle: predictions {pred_err:.2e}, loss {loss_err:.2e}, gradient tensors: median {med:.2e}, "
          f"90th percentile {p90:.2e}, 99th {errs[int(0.99 * len(errs))]:.2e}, worst {sig[0][0]:.2e} ({sig[0][1]}) of {len(errs)}")
    assert pred_err <= BF16_TRAIN_TOL['pred'] and loss_err <= BF16_TRAIN_TOL['loss']
    assert med <= BF16_TRAIN_TOL['grad_median'] and p90 <= BF16_TRAIN_TOL['grad_p90'] and sig[0][0] <= BF16_TRAIN_TOL['grad_max'], sig[:5]
    assert med > 1e-3                                      # (really the bf16 products: an f32 run sits at 3e-4 and below)


def _six_steps(bf16):
    from sgv3d_amd import hip_ops
    saved = hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS
    if bf16:
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
    try:
        B = 2
        model, bconf, hconf = _model(seed=1)
        model = model.cuda().train()
        model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
        imgs = synthetic.make_images(B, final=bconf['final_dim'], device='cuda', seed=4)
        mats = synthetic.make_mats(B, device='cuda')
        boxes, labels = _gt(B)
        targets = model.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
        opt = DataParallelAdamW(model.parameters(), lr=2e-4)
        losses = []
        for _ in range(6):
            opt.zero_grad()
            loss = model.loss(targets, model(imgs, mats))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses
    finally:
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = saved


def test_bf16_training_steps_track_the_f32_steps():
    """What the mixed-precision mode is for: the same six AdamW steps on the same data with f32 and with bf16 products.  The
    loss falls in both and the two trajectories stay within 2 % of each other at every step (f32 master weights, f32 AdamW)."""
    f32, bf16 = _six_steps(False), _six_steps(True)
    print("loss per step, f32 products :", [f"{v:.3f}" for v in f32])
    print("loss per step, bf16 products:", [f"{v:.3f}" for v in bf16])
    assert all(np.isfinite(bf16)) and bf16[-1] < bf16[0] and f32[-1] < f32[0]
    assert max(abs(a - b) / abs(a) for a, b in zip(f32, bf16)) <= 2e-2, (f32, bf16)


def test_training_steps_lower_the_loss_and_eval_sees_the_new_weights():
    B = 2
    model, bconf, hconf = _model(seed=1)
    model = model.cuda().train()
    model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
    imgs = synthetic.make_images(B, final=bconf['final_dim'], device='cuda', seed=4)
    mats = synthetic.make_mats(B, device='cuda')
    boxes, labels = _gt(B)
    targets = model.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    model.eval()
    with torch.no_grad():
        before = model(imgs, mats)[0][0]['heatmap'].clone()
    model.train()
    opt = DataParallelAdamW(model.parameters(), lr=2e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = model.loss(targets, model(imgs, mats))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        after = model(imgs, mats)[0][0]['heatmap']
    assert float((after - before).abs().max()) > 1e-4           # the packed inference weights were rebuilt
