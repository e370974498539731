#!/usr/bin/env python3
"""Golden fixtures of the rows either side of the hot path, produced by EXECUTING THE
REFERENCE'S OWN PYTHON in the build container (needs /root/reference, which never travels to the GPU box).

* losses.npz   ``losses.focal.FocalLoss`` (losses/focal.py:12-90, losses/_functional.py:37-108; pure torch, imported
               unmodified): loss values and input gradients for multiclass / binary / multilabel modes, with and
               without ``ignore_index``, 'mean' and 'sum'.

Outputs are DATA ONLY (inputs and expected outputs); no reference source text is stored.

    python tests/golden/make_golden_aux.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def make_losses():
    sys.path.insert(0, REF)
    from losses.focal import FocalLoss                        # the reference's class
    g = torch.Generator().manual_seed(7)
    out = {}
    cases = [
        ("mc_mean", dict(mode='multiclass', alpha=0.25, gamma=2, reduction="mean"), (3, 7, 6, 10)),     # the BSM config (:249)
        ("mc_sum_g15", dict(mode='multiclass', alpha=0.4, gamma=1.5, reduction="sum"), (2, 5, 4, 6)),
        ("mc_ignore", dict(mode='multiclass', alpha=None, gamma=2.0, ignore_index=3, reduction="mean"), (2, 4, 5, 7)),
        ("bin_mean", dict(mode='binary', alpha=0.25, gamma=2.0, reduction="mean"), (4, 1, 5, 6)),
        ("ml_sum_g1", dict(mode='multilabel', alpha=0.6, gamma=1.0, reduction="sum"), (2, 3, 4, 5)),
    ]
    for name, kw, shape in cases:
        x = (torch.randn(shape, generator=g, dtype=torch.float64) * 3).requires_grad_(True)
        if kw['mode'] == 'multiclass':
            y = torch.randint(0, shape[1], (shape[0],) + shape[2:], generator=g)
        else:
            y = torch.randint(0, 2, shape, generator=g)
        loss = FocalLoss(**kw)(x, y)
        loss.backward()
        out[name + "_x"] = x.detach().numpy()
        out[name + "_y"] = y.numpy()
        out[name + "_loss"] = np.float64(loss.item())
        out[name + "_grad"] = x.grad.numpy()
        out[name + "_kw"] = np.array(repr(kw))
        # the same inputs in float32 (what the harness feeds), value only
        out[name + "_loss32"] = np.float32(FocalLoss(**kw)(x.detach().float(), y).item())
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz:", sorted(k for k in out if k.endswith("_loss")))


if __name__ == "__main__":
    which = sys.argv[1:] or ["losses"]
    if "losses" in which:
        make_losses()
