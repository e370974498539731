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


def test_training_forward_backward_matches_oracle():
    B = 2
    model, bconf, hconf = _model()
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

    for t, (pl, rl) in enumerate(zip(preds, rpreds)):
        for k in pl[0]:
            got, want = pl[0][k].detach().cpu().double(), rl[0][k].detach()
            assert float((got - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max())), (t, k)
    assert abs(float(loss.detach()) - float(rloss.detach())) <= 1e-3 * abs(float(rloss.detach()))
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
    print('largest relative gradient errors:', [(f'{r:.1e}', n) for r, n, _ in worst[:5]])
    bad = [w for w in worst if w[0] > 5e-3 and w[2] > 1e-7]
    assert not bad, bad[:10]
    assert sorted(w[0] for w in worst)[len(worst) // 2] < 3e-4          # typical tensor: fp32 rounding only
    assert len(worst) > 150


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
