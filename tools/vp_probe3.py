#!/usr/bin/env python3
"""Dev probe: the planned gather (operator form and fused lift-splat form) on the cfg-2 and cfg-5 geometries, timed as a
hipGraph of 20 launches (GPU time, not the host's launch cadence).  Run once per kernel:
    python tools/vp_probe3.py                       # voxel-owner kernel (vp_gather_vox_kernel, default)
    SGV3D_VP_KERNEL=slot python tools/vp_probe3.py  # slot-balanced kernel of round 3
Also checks the fused form against lift + operator bit for bit, and the level-1 entry's time per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import _lib, synthetic as S              # noqa: E402
from sgv3d_amd.models.bev_height import BEVHeight       # noqa: E402


def graph_us(fn, reps=20):
    dev = torch.device("cuda")
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        g.replay()
        side.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            side.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    torch.cuda.current_stream(dev).wait_stream(side)
    return sorted(ts)[len(ts) // 2]


def main():
    kern = os.environ.get("SGV3D_VP_KERNEL", "vox")
    for name, conf, batch in (("cfg2", S.r50_256_conf, 1), ("cfg5", S.bsm_r101_256_conf, 1), ("cfg3_b4", S.r101_512_conf, 4)):
        bc, hc = conf()
        torch.manual_seed(0)
        model = BEVHeight(bc, hc).eval().cuda()
        bb = model.backbone
        mats = S.make_mats(batch, device="cuda")
        with torch.no_grad():
            geom, plan = bb.calibration(mats, 0)
        B, N = plan.B, plan.N
        C = bb.output_channels if not bc.get('is_bsm') else (bb.bev_channels + 3) // 4 * 4
        D = int(bb.height_channels)
        P = N // D
        X, Y, Z = bb._voxel_num_host
        g3 = geom.view(B, -1, 3).long()
        ok = (g3[..., 0] >= 0) & (g3[..., 0] < X) & (g3[..., 1] >= 0) & (g3[..., 1] < Y) & (g3[..., 2] >= 0) & (g3[..., 2] < Z)
        vid = (torch.arange(B, device="cuda").view(B, 1) * Y + g3[..., 1]) * X + g3[..., 0]
        cnt = torch.bincount(vid[ok], minlength=B * X * Y)
        ne = cnt[cnt > 0]
        edges = [0, 4, 8, 16, 32, 96, 10 ** 9]
        hist = [(int(((ne > lo) & (ne <= hi)).sum()), int(ne[(ne > lo) & (ne <= hi)].sum())) for lo, hi in zip(edges[:-1], edges[1:])]
        print(f"{name}: voxels {B * X * Y}, non-empty {ne.numel()}, kept points {int(ne.sum())}, mean {float(ne.float().mean()):.1f}, max {int(ne.max())}; "
              f"(voxels, points) with population 1-4 / 5-8 / 9-16 / 17-32 / 33-96 / >96: {hist}", flush=True)
        prob = torch.rand(B, D, P, device="cuda")
        ctx = torch.randn(B, P, C, device="cuda")
        lifted = (prob[..., None] * ctx[:, None]).reshape(B, N, C).contiguous()
        out = torch.empty(B, Y, X, C, device="cuda")
        want = plan.pool(lifted).clone()
        got = plan.lift_splat(prob, ctx).clone()
        same = bool(torch.equal(want, got))
        t_pool = graph_us(lambda: plan.pool(lifted, out=out))
        t_fused = graph_us(lambda: plan.lift_splat(prob, ctx, out=out))
        alg = 12.0 * B * N + 4.0 * B * N * C + 4.0 * B * Y * X * C
        alg_f = 4.0 * B * N + 4.0 * B * P * C + 4.0 * B * N + 4.0 * B * Y * X * C
        # level-1 entry (library-owned plan, compare + pos_memo + accumulate)
        lib = _lib.load()
        flat = geom.view(B, -1, 3)
        outz = torch.zeros(B, Y, X, C, device="cuda")
        pm = torch.full((B, N, 3), -1, dtype=torch.int32, device="cuda")
        l1 = lambda: lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, flat.data_ptr(), lifted.data_ptr(), outz.data_ptr(),
                                                     pm.data_ptr(), _lib.stream_handle(torch.device("cuda")))
        for _ in range(3):
            l1()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        evs[0].record()
        for r in range(20):
            l1()
            evs[r + 1].record()
        torch.cuda.synchronize()
        t_l1 = sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(20))[10] * 1e3
        t_l1g = graph_us(l1) if C % 4 == 0 else float('nan')
        print(f"{name} kernel={kern} N={N} C={C}: operator {t_pool:.1f} us = {alg / t_pool / 8e6:.3f} of 8 TB/s | fused {t_fused:.1f} us = "
              f"{alg_f / t_fused / 8e6:.3f} | level-1 {t_l1:.1f} us eager ({alg / t_l1 / 8e6:.3f}), {t_l1g:.1f} us in a graph "
              f"({alg / t_l1g / 8e6:.3f}) | fused == lift + operator bitwise: {same}", flush=True)
        del model, lifted, prob, ctx


if __name__ == "__main__":
    main()
