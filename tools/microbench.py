#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (development aid; bench.py is the judged benchmark).

    python tools/microbench.py [--out gpurun_out/microbench.json] [--what vp,conv,lift]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3   # median, min  (seconds)


def bench_vp(res):
    from oracle import geometry_ref as G   # dev tool only: synthetic cfg-2 geometry
    from sgv3d_amd import _lib
    from sgv3d_amd.ops.voxel_pooling import VoxelPlan
    lib = _lib.load()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    geo = np.load(os.path.join(ROOT, "tests", "golden", "geometry.npz"))
    n = "dair_p11_h5.5"
    for tag, bounds, C, D, fd, ds in (("cfg2_256", ([0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8]), 80, 90, (864, 1536), 16),
                                      ("cfg2_128", ([0, 102.4, 0.8], [-51.2, 51.2, 0.8], [-5, 3, 8]), 80, 90, (864, 1536), 16)):
        vs, vc, vn = G.voxel_params(*bounds)
        fr = G.create_frustum(fd, ds, [-2.0, 0.0, D])
        gi, _ = G.geom_xyz_for_camera(fr, geo[f"{n}/sensor2ego"], geo[f"{n}/sensor2virtual"], geo[f"{n}/intrin"],
                                      geo[f"{n}/ida"], geo[f"{n}/reference_height"], geo[f"{n}/bda"], vc, vs)
        N = gi.shape[0] * gi.shape[1] * gi.shape[2]
        X, Y, Z = (int(v) for v in vn)
        for kind in ("geometry", "uniform"):
            if kind == "uniform":
                g_np = np.random.default_rng(0).integers(-8, X + 8, size=(1, N, 3)).astype(np.int32)
                g_np[..., 2] = 0
            else:
                g_np = gi.reshape(1, N, 3)
            g = torch.from_numpy(g_np).cuda()
            f = torch.randn(1, N, C, device="cuda")
            out = torch.zeros(1, Y, X, C, device="cuda")
            st = _lib.stream_handle()
            alg_bytes = 12 * N + 4 * N * C + 4 * Y * X * C

            def atomic():
                out.zero_()
                lib.sgv3d_voxel_pooling_forward_atomic(1, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(), None, st)

            def atomic_nozero():
                lib.sgv3d_voxel_pooling_forward_atomic(1, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(), None, st)

            pm = torch.full((1, N, 3), -1, dtype=torch.int32, device="cuda")

            def level1():        # the symbol the reference's wrapper reaches (cached plan: compare + gated gather), with pos_memo
                lib.sgv3d_voxel_pooling_forward(1, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(), pm.data_ptr(), st)

            def level1_nomemo():
                lib.sgv3d_voxel_pooling_forward(1, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(), None, st)

            plan = VoxelPlan(g, (X, Y, Z))

            def planned():
                plan.pool(f, out)

            def build():
                VoxelPlan(g, (X, Y, Z))

            def build_nosort():
                VoxelPlan(g, (X, Y, Z), sort_segments=False)

            for name, fn in (("atomic+memset", atomic), ("atomic", atomic_nozero), ("planned_gather", planned),
                             ("level1_ext_entry", level1), ("level1_ext_entry_no_pos_memo", level1_nomemo),
                             ("plan_build", build), ("plan_build_nosort", build_nosort)):
                med, mn = timeit(fn)
                r = dict(case=f"{tag}/{kind}", kernel=name, us_median=med * 1e6, us_min=mn * 1e6)
                if name != "plan_build" and name != "plan_build_nosort":
                    r["alg_GBps"] = alg_bytes / med / 1e9
                    r["frac_of_8TBps"] = alg_bytes / med / 8e12
                res.append(r)
                print(r, flush=True)


def bench_lift(res):
    from sgv3d_amd.hip_ops import lift
    B, fH, fW, D, C = 1, 54, 96, 90, 80
    hc = torch.randn(B, fH, fW, D + C, device="cuda")
    med, mn = timeit(lambda: lift(hc, D, C))
    by = 4 * B * fH * fW * (D + C) + 4 * B * D * fH * fW * C
    r = dict(case="cfg2", kernel="lift", us_median=med * 1e6, us_min=mn * 1e6, GBps=by / med / 1e9)
    res.append(r)
    print(r, flush=True)


CONV_SHAPES = [
    # name, B, cin, H, W, cout, k, stride, pad, dil
    ("heightnet_3x3_512", 1, 512, 54, 96, 512, 3, 1, 1, 1),
    ("aspp_d6_512", 1, 512, 54, 96, 512, 3, 1, 6, 6),
    ("r50_l1_1x1_64_256", 1, 64, 216, 384, 256, 1, 1, 0, 1),
    ("r50_l1_3x3_64", 1, 64, 216, 384, 64, 3, 1, 1, 1),
    ("r50_l2_3x3_128", 1, 128, 108, 192, 128, 3, 1, 1, 1),
    ("r50_l3_3x3_256", 1, 256, 54, 96, 256, 3, 1, 1, 1),
    ("r50_l3_1x1_1024_256", 1, 1024, 54, 96, 256, 1, 1, 0, 1),
    ("r50_l4_3x3_512", 1, 512, 27, 48, 512, 3, 1, 1, 1),
    ("stem_7x7_4_64", 1, 4, 864, 1536, 64, 7, 2, 3, 1),
    ("head_stem_7x7_80_160", 1, 80, 256, 256, 160, 7, 2, 3, 1),
    ("head_l1_3x3_160", 1, 160, 128, 128, 160, 3, 1, 1, 1),
    ("head_shared_3x3_256_64", 1, 256, 256, 256, 64, 3, 1, 1, 1),
    ("head_branch1_3x3_64_2304", 1, 64, 256, 256, 2304, 3, 1, 1, 1),
]


def bench_conv(res):
    from sgv3d_amd.hip_ops import PackedConv
    for name, B, cin, H, W, cout, k, s, p, d in CONV_SHAPES:
        x = torch.randn(B, H, W, cin, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        for tile in (0, 1, 2, 3, 4):
            conv = PackedConv(w, stride=s, pad=p, dil=d, relu=True, tile=tile)
            oh, ow = conv.out_hw(H, W)
            out = torch.empty(B, oh, ow, cout, device="cuda")
            med, mn = timeit(lambda: conv(x, out), iters=10, warmup=2)
            flops = 2.0 * B * oh * ow * cout * k * k * cin
            r = dict(case=name, kernel=f"conv_tile{tile}", us_median=med * 1e6, us_min=mn * 1e6,
                     TFLOPs=flops / med / 1e12, frac_of_157TF=flops / med / 157.3e12)
            res.append(r)
            print(r, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "microbench.json"))
    ap.add_argument("--what", default="vp,lift,conv")
    args = ap.parse_args()
    res = []
    what = args.what.split(",")
    if "vp" in what:
        bench_vp(res)
    if "lift" in what:
        bench_lift(res)
    if "conv" in what:
        bench_conv(res)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
