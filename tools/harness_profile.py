#!/usr/bin/env python3
"""Dev probe: where a step of the reference harness's eval_step (sgv3d_amd/harness.py) spends its time on cfg-2 -- host wall
clock per phase with a device synchronisation after each (calibration .cuda(), forward, get_bboxes, .cpu().numpy())."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import harness as H, synthetic as S      # noqa: E402
from sgv3d_amd.models.bev_height import BEVHeight       # noqa: E402


def main():
    bc, hc = S.r50_256_conf()
    torch.manual_seed(0)
    model = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(model, 0, residual_gamma=0.3)
    model = model.cuda()
    imgs = S.make_images(1, bc['final_dim'], device='cuda', seed=0)
    host = S.make_mats(1, device='cpu')
    metas = [{'token': 'f0'}]
    sync = torch.cuda.synchronize
    with torch.no_grad():
        for _ in range(4):
            H.eval_step(model, H.make_batch(imgs, host))
        acc = {k: 0.0 for k in ("mats_cuda", "forward", "get_bboxes", "to_numpy", "whole_step")}
        n = 30
        for _ in range(n):
            sync(); t0 = time.perf_counter()
            mats = {k: v.cuda() for k, v in host.items()}
            sync(); t1 = time.perf_counter()
            preds = model(imgs, mats)
            sync(); t2 = time.perf_counter()
            res = model.get_bboxes(preds, metas)
            sync(); t3 = time.perf_counter()
            out = [[r[0].tensor.detach().cpu().numpy(), r[1].detach().cpu().numpy(), r[2].detach().cpu().numpy()] for r in res]
            t4 = time.perf_counter()
            acc["mats_cuda"] += t1 - t0; acc["forward"] += t2 - t1; acc["get_bboxes"] += t3 - t2; acc["to_numpy"] += t4 - t3
        sync(); t0 = time.perf_counter()
        for _ in range(n):
            H.eval_step(model, H.make_batch(imgs, host))
        sync(); acc["whole_step"] = time.perf_counter() - t0
    print({k: f"{v / n * 1e3:.3f} ms" for k, v in acc.items()}, "boxes", out[0][0].shape, flush=True)


if __name__ == "__main__":
    main()
