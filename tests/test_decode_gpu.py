"""GPU: box decode + circle NMS (csrc/decode.hip through BEVHeightHead.get_bboxes) against the numpy
restatement of mmdet3d's CenterHead.get_bboxes (oracle/decode_ref.py; parity unpinned)."""
import numpy as np
import pytest
import torch

from oracle import decode_ref
from sgv3d_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fake_preds(B, H, W, seed, n_obj=40):
    """Head-like maps: low background heat with Gaussian blobs (so that > max_num pixels compete)."""
    g = np.random.default_rng(seed)
    names = [('reg', 2), ('height', 1), ('dim', 3), ('rot', 2), ('vel', 2)]
    ncls = [1, 2, 2, 1, 2, 2]
    buf = g.standard_normal((B, 70, H, W)).astype(np.float32) * 0.3
    preds, off = [], 0
    yy, xx = np.mgrid[0:H, 0:W]
    for t, nc in enumerate(ncls):
        d = {}
        for n, c in names:
            d[n] = (off, c)
            off += c
        d['heatmap'] = (off, nc)
        hm = buf[:, off:off + nc]
        hm[:] = hm * 0.5 - 4.0
        for b in range(B):
            for _ in range(n_obj):
                c, y, x = g.integers(nc), g.integers(H), g.integers(W)
                hm[b, c] += 6.5 * np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / (2 * g.uniform(1.0, 6.0)))
        off += nc
        preds.append(d)
    return buf, preds


@pytest.mark.parametrize("H,W,B", [(256, 256, 2), (64, 96, 1)])
def test_get_bboxes_matches_oracle(H, W, B):
    from sgv3d_amd.layers.heads.bev_height_head import BEVHeightHead
    _, hc = S.r50_256_conf()
    head = BEVHeightHead(**hc)
    buf, layout = _fake_preds(B, H, W, seed=H)
    dbuf = torch.from_numpy(buf).to(DEV)
    preds_gpu = tuple([{k: dbuf[:, o:o + c] for k, (o, c) in d.items()}] for d in layout)
    preds_cpu = tuple([{k: buf[:, o:o + c] for k, (o, c) in d.items()}] for d in layout)
    res = head.get_bboxes(preds_gpu, img_metas=[dict() for _ in range(B)])
    ref = decode_ref.get_bboxes(preds_cpu, hc['bbox_coder'], hc['test_cfg'], head.num_classes)
    assert len(res) == B
    total = 0
    for i in range(B):
        boxes, scores, labels = res[i][0].tensor.cpu().numpy(), res[i][1].cpu().numpy(), res[i][2].cpu().numpy()
        rb, rs, rl = ref[i]
        assert boxes.shape == rb.shape and boxes.shape[1] == 9, (boxes.shape, rb.shape)
        assert np.array_equal(labels, rl)
        np.testing.assert_allclose(scores, rs, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(boxes, rb, rtol=1e-3, atol=1e-3)      # north_star: box regressions within 1e-3
        total += len(scores)
        # per task at most post_max_size survive, everything above the score threshold
        assert (scores > hc['bbox_coder']['score_threshold']).all()
    assert total > 20


def test_circle_nms_known_answer():
    """Three collinear centres 1 m apart, radius^2 = 1.5: the middle (2nd best) one is suppressed by the best."""
    dets = np.array([[0, 0, 0.9], [1, 0, 0.8], [2, 0, 0.7], [10, 10, 0.95]], np.float32)
    keep = decode_ref.circle_nms(dets, 1.5, 83)
    assert list(keep) == [3, 0, 2]


def test_full_model_decode_runs():
    """End to end: forward + get_bboxes on the small config, compared with the oracle on the HIP maps."""
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.small_conf(depth=18)
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(m, 1)
    with torch.no_grad():                                    # make some heat rise above the 0.1 threshold
        for th in m.head.task_heads:
            th.heatmap[1].bias.fill_(-1.0)
            th.heatmap[1].weight.mul_(3.0)
    m = m.to(DEV)
    imgs, mats = S.make_images(2, bc['final_dim'], device=DEV, seed=4), S.make_mats(2, device=DEV, scale=128 / 864)
    with torch.no_grad():
        preds = m(imgs, mats)
        res = m.get_bboxes(preds, [dict(), dict()])
    preds_cpu = tuple([{k: v.cpu().numpy() for k, v in p[0].items()}] for p in preds)
    ref = decode_ref.get_bboxes(preds_cpu, hc['bbox_coder'], hc['test_cfg'], m.head.num_classes)
    for i in range(2):
        assert res[i][0].tensor.shape == ref[i][0].shape
        np.testing.assert_allclose(res[i][0].tensor.cpu().numpy(), ref[i][0], rtol=1e-3, atol=1e-3)
        assert np.array_equal(res[i][2].cpu().numpy(), ref[i][2])
