"""CPU: the committed per-layer choices (tune/gfx950_*.json) only name algorithms this build knows."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_tune_dbs_name_known_tiles():
    from sgv3d_amd import hip_ops
    files = sorted(glob.glob(os.path.join(ROOT, "tune", "gfx950_*.json")))
    assert files, "no committed tune DBs"
    known = set(hip_ops.TILE_NAMES)
    for f in files:
        db = json.load(open(f))
        assert db, f
        for sig, choice in db.items():
            assert isinstance(choice, list) and len(choice) == 2, (f, sig, choice)
            t, s = int(choice[0]), int(choice[1])
            if sig.startswith("centerhead_branches"):
                assert t in (100, 101, 102), (f, sig, choice)            # fused F(2x2) / two kernels / fused F(4x4)
            elif sig.startswith("wgrad|"):
                assert t >= 0 and s >= 0, (f, sig, choice)               # weight-gradient (tile, pixel split); 0 = the kernel's own rule
                if sig.endswith("xbf16"):
                    assert t in (0, 1, 4, 6), (f, sig, choice)           # bf16 weight gradients: 64x64 / 128x128 per tap, 6 = all taps (3x3)
            elif sig.startswith("pair|"):
                assert t in (0, 1), (f, sig, choice)                     # fused conv2 + conv3 launch or not
            else:
                assert t in known and s >= 1, (f, sig, choice)
                if "bf16" not in sig:
                    assert t < 30 or t >= 40, (f, sig, choice)           # 31..39 are bf16-only kernels
                else:
                    assert t not in (40, 44, 45, 46, 47) and t not in (5, 6, 8, 9, 10, 15), (f, sig, choice)   # f32-only algorithms


def test_every_candidate_tile_has_a_name():
    from sgv3d_amd import hip_ops
    for t in (hip_ops.TILE_WINO, hip_ops.TILE_WINO_RES, hip_ops.TILE_PATCH, hip_ops.TILE_WINO_HALF, hip_ops.TILE_F4RES) \
            + hip_ops.WINO4_TILES + hip_ops.DW_TILES + hip_ops.OCC5_TILES + (1, 2, 3, 4, 21, 22, 23, 24):
        assert t in hip_ops.TILE_NAMES, t
