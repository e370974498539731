"""sgv3d_amd/input_contract.py (the mats_dict producer) against outputs of the reference's
own dataset helpers captured in tests/golden/input_contract.npz (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from sgv3d_amd import input_contract as IC
from sgv3d_amd import synthetic as S

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_contract.npz"))


def test_plane_denorm_virtual_and_height_match_reference():
    assert np.allclose(IC.equation_plane(G["plane_points"]), G["plane"], rtol=0, atol=1e-12)
    for i, s2e in enumerate(G["sensor2ego"]):
        e2s = np.linalg.inv(s2e.astype(np.float64))
        d = IC.get_denorm(e2s)
        assert np.allclose(d, G["denorm"][i], rtol=0, atol=1e-12)
        s2v = IC.get_sensor2virtual(d)
        assert s2v.dtype == np.float32 and np.allclose(s2v, G["sensor2virtual"][i], rtol=0, atol=2e-7)
        h = IC.get_reference_height(d)
        assert h.dtype == np.float32 and h == G["reference_height"][i]


def test_ida_sampling_and_matrix_match_reference():
    for key, src, final, bot in (("dair", (1080, 1920), (864, 1536), (0.0, 0.0)), ("nusc", (900, 1600), (256, 704), (0.0, 0.22))):
        resize, dims, crop, flip, rot = IC.ida_resize_crop(src, final, bot)
        got = np.array([resize, *dims, *crop, float(flip), float(rot)])
        assert np.allclose(got, G[f"ida_{key}_sample"], rtol=0, atol=1e-12), key
        assert np.allclose(IC.ida_matrix(resize, crop, flip, rot), G[f"ida_{key}_mat"], rtol=0, atol=1e-6), key
    a = G["ida_aug_args"]
    m = IC.ida_matrix(a[0], tuple(a[1:5]), bool(a[5]), a[6])
    assert np.allclose(m, G["ida_aug_mat"], rtol=1e-5, atol=1e-4)
    # the DAIR evaluation matrix is what the synthetic bench uses: diag(0.8, 0.8, 1, 1), no translation
    assert np.allclose(G["ida_dair_mat"], np.diag([0.8, 0.8, 1, 1]), atol=1e-7)


def test_bda_matrix_matches_reference():
    assert np.allclose(IC.bda_matrix()[:3, :3], G["bda_identity"], atol=0)
    r, s, fx, fy = G["bda_aug_args"]
    assert np.allclose(IC.bda_matrix(r, s, bool(fx), bool(fy))[:3, :3], G["bda_aug"], rtol=0, atol=1e-6)


def test_collate_layout_and_synthetic_calibration_agree():
    """collate_mats builds the tensors BEVHeight.forward takes; the synthetic calibrations of the bench /
    tests are produced by the same helpers."""
    cams = [S.make_calib(), S.make_calib(pitch_deg=14.0, cam_h=6.1, yaw_deg=2.0, roll_deg=0.4)]
    mats = IC.collate_mats([dict(sensor2ego=c["sensor2ego"], intrin=c["intrin"], ida=c["ida"], bda=c["bda"]) for c in cams])
    assert mats["sensor2ego_mats"].shape == (2, 1, 1, 4, 4) and mats["reference_heights"].shape == (2, 1, 1)
    assert mats["bda_mat"].shape == (2, 4, 4) and all(v.dtype == torch.float32 for v in mats.values())
    for i, c in enumerate(cams):
        assert np.allclose(mats["sensor2virtual_mats"][i, 0, 0].numpy(), c["sensor2virtual"], atol=1e-6)
        assert abs(float(mats["reference_heights"][i]) - float(c["reference_height"])) < 1e-5
