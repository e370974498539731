"""Training-side head functions on the MI355X against the oracle: target assignment
(indices / masks bit-exact, Gaussian values and box codes to fp32 rounding) and the detection loss with its
gradient (against torch autograd in float64)."""
import numpy as np
import pytest
import torch

from oracle import train_head_ref as R
from sgv3d_amd import synthetic
from sgv3d_amd.layers.heads.bev_height_head import BEVHeightHead

pytestmark = pytest.mark.gpu

KEYS = (('heatmap', None), ('reg', 2), ('height', 1), ('dim', 3), ('rot', 2), ('vel', 2))


@pytest.fixture(scope="module")
def head():
    _, head_conf = synthetic.r50_256_conf()
    return BEVHeightHead(**head_conf).cuda().eval()


def _names(head):
    return head.class_names


def _check_targets(head, boxes, labels, cfg=None):
    old = head.train_cfg
    if cfg is not None:
        head.train_cfg = cfg
    try:
        got = head.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    finally:
        head.train_cfg = old
    want = R.get_targets([b.numpy() for b in boxes], [l.numpy() for l in labels], _names(head), cfg or old)
    for t in range(len(_names(head))):
        assert torch.equal(got[2][t].cpu(), torch.from_numpy(want[2][t])), f"ind of task {t}"
        assert torch.equal(got[3][t].cpu(), torch.from_numpy(want[3][t])), f"mask of task {t}"
        np.testing.assert_allclose(got[1][t].cpu().numpy(), want[1][t], rtol=2e-6, atol=2e-7, err_msg=f"anno_box {t}")
        h_got, h_want = got[0][t].cpu().numpy(), want[0][t]
        assert np.array_equal(h_got == 1.0, h_want == 1.0), "peaks"
        assert np.array_equal(h_got == 0.0, h_want == 0.0), "support of the Gaussians"
        assert np.abs(h_got - h_want).max() <= 6e-8          # float64 exp on both sides, one fp32 rounding
        assert (h_got != h_want).mean() < 1e-4
    return got, want


@pytest.mark.parametrize("seed", range(6))
def test_targets_match_oracle(head, seed):
    boxes, labels = synthetic.make_gt(3, seed=seed, n_range=(0, 60))
    got, want = _check_targets(head, boxes, labels)
    assert sum(int(m.sum()) for m in got[3]) > 10


def test_targets_empty_and_overflowing_samples(head):
    boxes, labels = synthetic.make_gt(2, seed=11, n_range=(30, 30), stress=False)
    boxes[0], labels[0] = boxes[0][:0], labels[0][:0]                      # a sample without boxes
    labels[1][:] = 0                                                       # 30 cars into 8 slots
    cfg = dict(head.train_cfg, max_objs=8)
    got, want = _check_targets(head, boxes, labels, cfg)
    assert int(got[3][0][1].sum()) == 8 and int(got[3][0][0].sum()) == 0


def test_targets_other_grid(head):
    cfg = dict(head.train_cfg, grid_size=[512, 512, 1], voxel_size=[0.2, 0.2, 8], gaussian_overlap=0.5, min_radius=1)
    boxes, labels = synthetic.make_gt(2, seed=5)
    _check_targets(head, boxes, labels, cfg)


def _preds(names, B, H, W, seed, device, requires_grad=False, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    total = sum(len(n) + 10 for n in names)
    buf = (torch.randn(B, total, H, W, generator=g) * scale).to(device)
    if requires_grad:
        buf.requires_grad_(True)
    preds, c0 = [], 0
    for n in names:
        d = {}
        for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('rot', 2), ('vel', 2), ('heatmap', len(n))):
            d[k] = buf[:, c0:c0 + c]
            c0 += c
        preds.append([d])
    return buf, preds


def _torch_loss(targets, preds, code_weights, box_w):
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
        flat = anno.permute(0, 2, 3, 1).reshape(B, H * W, C)
        pred = flat.gather(1, targets[2][t][:, :, None].expand(-1, -1, C))
        m = targets[3][t].double()[:, :, None] * torch.tensor(code_weights, dtype=torch.float64)
        num = max(float(targets[3][t].sum()), 1e-4)
        total = total + ((pred - targets[1][t]).abs() * m).sum() / num * box_w
    return total


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (1, 4.0)])
def test_loss_and_gradient_match_autograd(head, seed, scale):
    boxes, labels = synthetic.make_gt(2, seed=seed, n_range=(20, 60))
    targets = head.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    buf, preds = _preds(_names(head), 2, 256, 256, seed, 'cuda', requires_grad=True, scale=scale)
    loss = head.loss(targets, preds)
    loss.backward()
    # float64 autograd reference on the CPU
    tc = tuple([x.cpu().double() if x.dtype == torch.float32 else x.cpu() for x in part] for part in targets)
    rbuf, rpreds = _preds(_names(head), 2, 256, 256, seed, 'cpu', scale=scale)
    rbuf = rbuf.double().requires_grad_(True)
    c0 = 0
    for pl, n in zip(rpreds, _names(head)):
        for k, c in (('reg', 2), ('height', 1), ('dim', 3), ('rot', 2), ('vel', 2), ('heatmap', len(n))):
            pl[0][k] = rbuf[:, c0:c0 + c]
            c0 += c
    want = _torch_loss(tc, rpreds, head.train_cfg['code_weights'], 0.25)
    want.backward()
    assert abs(float(loss.detach()) - float(want.detach())) <= 2e-5 * abs(float(want.detach()))
    g_got, g_want = buf.grad.cpu().double(), rbuf.grad
    scale_g = float(g_want.abs().max())
    # cells whose sigmoid sits on a clamp bound switch their gradient on or off with the last bit of the sigmoid
    sg = torch.sigmoid(rbuf.detach())
    edge = ((sg - 1e-4).abs() < 1e-9) | ((sg - (1 - 1e-4)).abs() < 3e-7)
    assert int(edge.sum()) < 2000
    assert float(((g_got - g_want).abs() * (~edge)).max()) <= 2e-5 * scale_g
    # several boxes on one cell: their L1 gradients must have been summed, not overwritten
    shared = 0
    for t in range(len(_names(head))):
        for b in range(2):
            live = targets[2][t][b][targets[3][t][b].bool()]
            shared += int(live.numel() - live.unique().numel())
    assert shared > 0
    # the oracle (numpy float64) agrees as well
    o_total, _ = R.loss(tuple([x.cpu().numpy() for x in part] for part in targets),
                        [{k: v.detach().cpu().numpy() for k, v in pl[0].items()} for pl in preds],
                        head.train_cfg['code_weights'], 0.25)
    assert abs(float(loss.detach()) - o_total) <= 2e-5 * abs(o_total)


def test_loss_is_deterministic_and_handles_no_boxes(head):
    boxes, labels = synthetic.make_gt(2, seed=4)
    targets = head.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    _, preds = _preds(_names(head), 2, 256, 256, 7, 'cuda')
    a = head.loss(targets, preds)
    ga = [g['heatmap'].clone() for g in head.last_pred_grads]
    b = head.loss(targets, preds)
    assert torch.equal(a, b) and all(torch.equal(x, g['heatmap']) for x, g in zip(ga, head.last_pred_grads))
    empty = head.get_targets([b[:0].cuda() for b in boxes], [l[:0].cuda() for l in labels])
    l0 = head.loss(empty, preds)
    assert torch.isfinite(l0) and float(l0) > 0
    assert all(float(g[k].abs().sum()) == 0 for g in head.last_pred_grads for k in ('reg', 'height', 'dim', 'rot', 'vel'))


def test_separate_prediction_tensors(head):
    """Prediction maps that are independent tensors (the layout mmdet3d's CenterHead returns) give the same loss."""
    boxes, labels = synthetic.make_gt(1, seed=9, n_range=(10, 30))
    targets = head.get_targets([b.cuda() for b in boxes], [l.cuda() for l in labels])
    _, preds = _preds(_names(head), 1, 256, 256, 3, 'cuda')
    a = head.loss(targets, preds)
    sep = [[{k: v.clone() for k, v in pl[0].items()}] for pl in preds]
    b = head.loss(targets, sep)
    assert torch.equal(a, b)
