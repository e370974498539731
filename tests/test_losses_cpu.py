"""The focal-loss oracle (oracle/focal_ref.py) against the reference's own ``losses.focal.FocalLoss`` outputs
(tests/golden/losses.npz, produced by tests/golden/make_golden_aux.py), and the label down-sampling restatement."""
import ast
import os

import numpy as np
import torch

from oracle import focal_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
CASES = sorted(k[:-5] for k in GOLD.files if k.endswith("_loss"))


def test_golden_has_every_mode():
    assert CASES == ['bin_mean', 'mc_ignore', 'mc_mean', 'mc_sum_g15', 'ml_sum_g1']


def test_oracle_reproduces_reference_values_and_gradients():
    for name in CASES:
        kw = ast.literal_eval(str(GOLD[name + "_kw"]))
        x = torch.from_numpy(GOLD[name + "_x"]).requires_grad_(True)
        y = torch.from_numpy(GOLD[name + "_y"])
        loss = R.focal_loss(x, y, kw['mode'], kw.get('alpha'), kw.get('gamma', 2.0), kw.get('ignore_index'), kw['reduction'])
        loss.backward()
        assert abs(float(loss.detach()) - float(GOLD[name + "_loss"])) <= 1e-12 * max(1.0, abs(float(GOLD[name + "_loss"]))), name
        np.testing.assert_allclose(x.grad.numpy(), GOLD[name + "_grad"], rtol=1e-11, atol=1e-14, err_msg=name)
        l32 = R.focal_loss(x.detach().float(), y, kw['mode'], kw.get('alpha'), kw.get('gamma', 2.0), kw.get('ignore_index'),
                           kw['reduction'])
        assert abs(float(l32) - float(GOLD[name + "_loss32"])) <= 2e-6 * abs(float(GOLD[name + "_loss32"])), name


def test_label_downsampling_takes_the_block_maximum():
    g = torch.Generator().manual_seed(0)
    gt = torch.randint(0, 7, (2, 1, 32, 48), generator=g, dtype=torch.uint8)
    lab = R.downsample_gt_semantic(gt, 8)
    assert lab.shape == (2, 4, 6) and lab.dtype == torch.int64
    for b in range(2):
        for i in range(4):
            for j in range(6):
                assert int(lab[b, i, j]) == int(gt[b, 0, 8 * i:8 * i + 8, 8 * j:8 * j + 8].max())
