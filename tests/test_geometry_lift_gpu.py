"""GPU: geometry (voxel indices bit-exact) and lift kernels against the oracle / golden vectors."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import geometry_ref as G
from oracle import lift_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CALIBS = ["dair_p11_h5.5", "p5_h8_yaw3", "p20_h4_roll2", "p14_h6.3_yaw-7_roll-1"]
BOUNDS256 = ([0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
BOUNDS128 = ([0, 102.4, 0.8], [-51.2, 51.2, 0.8], [-5, 3, 8])


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV)


def _run(hip, frustum, cams, bounds, prep_override=None, want_float=True):
    """cams: list of dicts (one per camera, batch = len(cams), 1 cam per batch element)."""
    lib = hip.load()
    n = len(cams)
    D, fH, fW, _ = frustum.shape
    vs, vc, vn = G.voxel_params(*bounds)
    mats = {k: _t(np.stack([c[k] for c in cams])) for k in ("sensor2ego", "sensor2virtual", "intrin", "ida")}
    prep = torch.empty(n, 3, 4, 4, device=DEV)
    st = hip.stream_handle()
    hip.check(lib.sgv3d_calib_prep(n, mats["sensor2ego"].data_ptr(), mats["sensor2virtual"].data_ptr(),
                                   mats["intrin"].data_ptr(), mats["ida"].data_ptr(), prep.data_ptr(), st), "prep")
    if prep_override is not None:
        prep = _t(prep_override)
    fr = _t(frustum)
    refh = _t(np.array([c["reference_height"] for c in cams], np.float32))
    bda = _t(np.stack([c["bda"] for c in cams]))
    gi = torch.empty(n, D, fH, fW, 3, dtype=torch.int32, device=DEV)
    gf = torch.empty(n, D, fH, fW, 3, device=DEV) if want_float else None
    import ctypes
    vcp = (ctypes.c_float * 3)(*[float(v) for v in vc])
    vsp = (ctypes.c_float * 3)(*[float(v) for v in vs])
    hip.check(lib.sgv3d_geometry_voxel_index(n, 1, D, fH, fW, fr.data_ptr(), prep.data_ptr(), refh.data_ptr(),
                                             bda.data_ptr(), vcp, vsp, gi.data_ptr(),
                                             gf.data_ptr() if gf is not None else None, st), "geometry")
    torch.cuda.synchronize()
    return gi.cpu().numpy(), (gf.cpu().numpy() if gf is not None else None), prep.cpu().numpy()


def _calib(geo, n):
    c = {k: geo[f"{n}/{k}"] for k in ("sensor2ego", "sensor2virtual", "intrin", "ida", "bda")}
    c["reference_height"] = float(geo[f"{n}/reference_height"])
    return c


def test_calib_prep_matches_oracle_bitwise(hip, golden):
    geo = golden["geometry"]
    cams = [_calib(geo, n) for n in CALIBS + ["nan_ray_small"]]
    small = G.create_frustum((80, 112), 16, [-2.0, 0.0, 6])
    _, _, prep = _run(hip, small, cams, BOUNDS256)
    for i, c in enumerate(cams):
        ida_inv, cv, ce = G.calib_prep(c["sensor2ego"], c["sensor2virtual"], c["intrin"], c["ida"])
        assert np.array_equal(prep[i, 0], ida_inv) and np.array_equal(prep[i, 1], cv) and np.array_equal(prep[i, 2], ce)
        # and close to the platform inverse the reference would have used
        np.testing.assert_allclose(prep[i, 2], c["sensor2ego"] @ np.linalg.inv(c["sensor2virtual"].astype(np.float64)),
                                   rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", CALIBS + ["nan_ray_small"])
def test_small_golden_bit_exact(hip, golden, name):
    """Reduced-size golden tensors captured from the reference: float bits and int32 indices equal,
    both with the reference's own 4x4 products and with the device-side preparation."""
    geo = golden["geometry"]
    c = _calib(geo, name)
    small = G.create_frustum((80, 112), 16, [-2.0, 0.0, 6])
    ref_prep = np.stack([geo[f"{name}/ref_ida_inv"], geo[f"{name}/ref_combine_virtual"], geo[f"{name}/ref_combine_ego"]])[None]
    gi, gf, _ = _run(hip, small, [c], BOUNDS256, prep_override=ref_prep)
    ref = geo[f"{name}/small/geom"]
    same = (gf[0].view(np.int32) == ref.view(np.int32)) | (np.isnan(gf[0]) & np.isnan(ref))
    assert same.all()
    assert np.array_equal(gi[0], geo[f"{name}/small/geom_xyz"])          # incl. NaN -> 0, inf saturation
    gi2, _, _ = _run(hip, small, [c], BOUNDS256)
    assert np.array_equal(gi2[0], geo[f"{name}/small/geom_xyz"])


@pytest.mark.parametrize("tag,bounds", [("full256", BOUNDS256), ("full128", BOUNDS128)])
def test_full_size_hash_batch4(hip, golden, tag, bounds):
    """cfg-2 size, four cameras in one launch: every int32 index tensor hashes to the reference's."""
    geo = golden["geometry"]
    cams = [_calib(geo, n) for n in CALIBS]
    full = G.create_frustum((864, 1536), 16, [-2.0, 0.0, 90])
    gi, _, _ = _run(hip, full, cams, bounds, want_float=False)
    for i, n in enumerate(CALIBS):
        assert np.array_equal(gi[i][::7, ::5, ::9], geo[f"{n}/{tag}/geom_xyz_sample"])
        assert hashlib.sha256(gi[i].tobytes()).digest() == geo[f"{n}/{tag}/geom_xyz_sha256"].tobytes(), n


BIG_CFGS = {"cfg3_512": ((1088, 1920), 16, [-2.0, 0.0, 90], ([0, 102.4, 0.2], [-51.2, 51.2, 0.2], [-5, 3, 8])),
            "cfg5_s8d180": ((864, 1536), 8, [-2.0, 3.5, 180], BOUNDS256)}


@pytest.mark.parametrize("tag", list(BIG_CFGS))
def test_cfg3_cfg5_full_size_hash(hip, golden, tag):
    """BASELINE cfg-3 (1088x1920 frame, 0.2 m cells -> 512x512 BEV, 734 400 points per camera) and cfg-5 (SGV3D BSM:
    stride-8 frustum, 180 height bins, 3 732 480 points per camera) at full size, five calibrations in one launch.
    Fed with the reference's own 4x4 products, every int32 index tensor hashes to the one the reference's get_geometry +
    quantise produced; with the device-side 4x4 preparation the kernel equals the oracle bit for bit (which differs from
    the reference tensor on <= 20 border points of one calibration, tests/test_oracle_cpu.py explains why)."""
    geo = golden["geometry"]
    names = CALIBS + ["p11_h5.5_fullres"]
    cams = [_calib(geo, n) for n in names]
    fd, ds, db, bounds = BIG_CFGS[tag]
    fr = G.create_frustum(fd, ds, db)
    ref_prep = np.stack([np.stack([geo[f"{n}/ref_ida_inv"], geo[f"{n}/ref_combine_virtual"], geo[f"{n}/ref_combine_ego"]])
                         for n in names])
    gi, _, _ = _run(hip, fr, cams, bounds, prep_override=ref_prep, want_float=False)
    for i, n in enumerate(names):
        assert np.array_equal(gi[i][::7, ::5, ::9], geo[f"{n}/{tag}/geom_xyz_sample"])
        assert hashlib.sha256(gi[i].tobytes()).digest() == geo[f"{n}/{tag}/geom_xyz_sha256"].tobytes(), n
    own, _, _ = _run(hip, fr, cams, bounds, want_float=False)
    vs, vc, vn = G.voxel_params(*bounds)
    for i, (n, c) in enumerate(zip(names, cams)):
        ora, _ = G.geom_xyz_for_camera(fr, c["sensor2ego"], c["sensor2virtual"], c["intrin"], c["ida"],
                                       c["reference_height"], c["bda"], vc, vs)
        assert np.array_equal(own[i], ora), n
        moved = (np.abs(own[i].astype(np.int64) - gi[i]).sum(-1) > 0).sum()
        assert moved <= 20 and (moved == 0 or n == "p14_h6.3_yaw-7_roll-1"), (n, int(moved))


def test_lift_golden(hip, golden):
    lib = hip.load()
    lf = golden["lift"]
    B, D, C, fH, fW = (int(v) for v in lf["dims"])
    P = fH * fW
    hc = _t(lf["height_feature"].transpose(0, 2, 3, 1).reshape(B, P, D + C))      # channel-last
    prob = torch.empty(B, D, P, device=DEV)
    lifted = torch.empty(B, D, P, C, device=DEV)
    hip.check(lib.sgv3d_lift(B, P, D, C, hc.data_ptr(), prob.data_ptr(), lifted.data_ptr(), hip.stream_handle()), "lift")
    torch.cuda.synchronize()
    np.testing.assert_allclose(lifted.cpu().numpy().reshape(lf["lifted"].shape), lf["lifted"], rtol=1e-5, atol=1e-6)
    p_ref, _ = lift_ref.lift(lf["height_feature"], D, C)
    np.testing.assert_allclose(prob.cpu().numpy().reshape(B, D, fH, fW), p_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("D,C,P", [(90, 80, 54 * 96), (180, 87, 1000), (7, 3, 33)])
def test_lift_vs_oracle(hip, D, C, P):
    lib = hip.load()
    rng = np.random.default_rng(D)
    B = 2
    hf = (rng.standard_normal((B, D + C, 1, P)) * 3).astype(np.float32)
    hc = _t(hf.transpose(0, 2, 3, 1).reshape(B, P, D + C))
    prob = torch.empty(B, D, P, device=DEV)
    lifted = torch.empty(B, D, P, C, device=DEV)
    hip.check(lib.sgv3d_lift(B, P, D, C, hc.data_ptr(), prob.data_ptr(), lifted.data_ptr(), hip.stream_handle()), "lift")
    torch.cuda.synchronize()
    p_ref, l_ref = lift_ref.lift(hf, D, C)
    np.testing.assert_allclose(prob.cpu().numpy(), p_ref.reshape(B, D, P), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lifted.cpu().numpy(), l_ref.reshape(B, D, P, C), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(prob.sum(1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_random_calibration_fuzz_bit_exact(hip, seed):
    """Random roadside poses (pitch 2..35 deg, height 3..12 m, yaw, roll), intrinsics, resize / crop and BEV
    grids: voxel indices of the HIP kernels == the oracle restatement, every point, every bit."""
    from sgv3d_amd import synthetic as S
    rng = np.random.default_rng(500 + seed)
    cams = []
    for _ in range(3):
        c = S.make_calib(pitch_deg=float(rng.uniform(2, 35)), cam_h=float(rng.uniform(3, 12)),
                         yaw_deg=float(rng.uniform(-15, 15)), roll_deg=float(rng.uniform(-3, 3)),
                         fx=float(rng.uniform(900, 2600)), fy=float(rng.uniform(900, 2600)),
                         cx=float(rng.uniform(700, 1200)), cy=float(rng.uniform(400, 700)),
                         resize=float(rng.uniform(0.4, 1.0)), crop=(float(rng.integers(0, 40)), float(rng.integers(0, 40))))
        cams.append({k: c[k] for k in ("sensor2ego", "sensor2virtual", "intrin", "ida", "bda")} | {"reference_height": float(c["reference_height"])})
    step = float(rng.choice([0.2, 0.4, 0.8]))
    half = float(rng.choice([25.6, 51.2]))
    bounds = ([0, 2 * half, step], [-half, half, step], [-5, 3, 8])
    D = int(rng.choice([6, 12, 20]))
    frustum = G.create_frustum((96, 160), 16, [float(rng.uniform(-3, -0.5)), float(rng.uniform(0, 3.5)), D])
    gi, _, prep = _run(hip, frustum, cams, bounds, want_float=False)
    vs, vc, vn = G.voxel_params(*bounds)
    for i, c in enumerate(cams):
        ref, _ = G.geom_xyz_for_camera(frustum, c["sensor2ego"], c["sensor2virtual"], c["intrin"], c["ida"],
                                       c["reference_height"], c["bda"], vc, vs)
        assert np.array_equal(gi[i], ref), (seed, i, int((gi[i] != ref).sum()))
