#!/usr/bin/env python3
"""Dev probe: the level-1 entry (sgv3d_voxel_pooling_forward) inside a hipGraph when geom_xyz CHANGES with every call: each
call's one-launch rebuild (vp_plan_build_one_kernel) runs for real.  Against the multi-launch build of VoxelPlan."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import _lib, synthetic as S                          # noqa: E402
from sgv3d_amd.models.bev_height import BEVHeight                   # noqa: E402
from sgv3d_amd.ops.voxel_pooling import VoxelPlan                   # noqa: E402
from tools.vp_probe3 import graph_us                                # noqa: E402


def main():
    dev = torch.device("cuda")
    bc, hc = S.r50_256_conf()
    m = BEVHeight(bc, hc).eval().to(dev)
    two = S.make_mats(2, device=dev)
    geoms = []
    with torch.no_grad():
        for i in range(2):
            g, _ = m.backbone.calibration({k: v[i:i + 1].clone() for k, v in two.items()}, 0)
            geoms.append(g.reshape(1, -1, 3).clone())
    B, N, C, X, Y, Z = 1, geoms[0].shape[1], 80, 256, 256, 1
    feats = torch.randn(B, N, C, device=dev)
    out = torch.zeros(B, Y, X, C, device=dev)
    lib = _lib.load()
    k = [0]

    def changing():
        k[0] ^= 1
        _lib.check(lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, geoms[k[0]].data_ptr(), feats.data_ptr(), out.data_ptr(), None,
                                                   _lib.stream_handle(dev)), "level1")

    def same():
        _lib.check(lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, geoms[0].data_ptr(), feats.data_ptr(), out.data_ptr(), None,
                                                   _lib.stream_handle(dev)), "level1")
    t_same = graph_us(same, reps=10)
    t_chg = graph_us(changing, reps=10)
    t_build = graph_us(lambda: VoxelPlan(geoms[0], (X, Y, Z), cached=False), reps=5)
    print(f"cfg-2 level-1 in a graph: unchanged geom_xyz {t_same:.1f} us per call | geom_xyz alternating between two cameras {t_chg:.1f} us "
          f"(one-launch rebuild = {t_chg - t_same:.1f} us on top) | VoxelPlan multi-launch build alone {t_build:.1f} us")


if __name__ == "__main__":
    main()
