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


def test_info_record_adapter():
    """info entry in the layout scripts/gen_info_dair.py:111-196 writes -> mats_dict / ground-truth boxes."""
    from sgv3d_amd import input_contract as IC
    from sgv3d_amd import synthetic as S
    calib = S.make_calib()
    s2e = calib['sensor2ego'].astype(np.float64)
    cam_info = dict(height=1080, width=1920, filename='image/000001.jpg',
                    ego_pose=dict(translation=[0.0, 0.0, 0.0], rotation=[1.0, 0.0, 0.0, 0.0]),
                    calibrated_sensor=dict(translation=s2e[:3, 3], rotation_matrix=s2e[:3, :3], camera_intrinsic=calib['intrin'][:3, :3]))
    yaw = 0.7
    ann = dict(category_name='car', translation=np.array([30.0, -4.0, -1.2]), size=np.array([1.9, 4.4, 1.6]),
               rotation=[np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)], yaw_lidar=yaw, num_lidar_pts=3, num_radar_pts=0,
               velocity=np.zeros(3))
    skipped = dict(ann, category_name='animal')
    info = dict(sample_token='image/000001.jpg', cam_infos={'CAM_FRONT': cam_info}, ann_infos=[ann, skipped], sweeps=[])
    mats = IC.mats_from_infos([info], final_dim=(864, 1536))
    ref = S.make_mats(1)                                          # the same camera through the direct producer
    for k in ref:
        assert mats[k].shape == ref[k].shape, k
        assert torch.allclose(mats[k], ref[k], rtol=1e-5, atol=1e-5), k
    boxes, labels = IC.gt_from_info(info, ['car', 'truck', 'pedestrian'])
    assert boxes.shape == (1, 9) and labels.tolist() == [0]
    assert torch.allclose(boxes[0], torch.tensor([30.0, -4.0, -1.2, 4.4, 1.9, 1.6, yaw, 0.0, 0.0]), atol=1e-6)
    # a quaternion instead of the rotation matrix, and a moved / turned ego pose
    q = [np.cos(0.25), 0.0, 0.0, np.sin(0.25)]                    # yaw 0.5
    cam_q = dict(cam_info, ego_pose=dict(translation=[1.0, 2.0, 0.0], rotation=q))
    b2, _ = IC.gt_from_info(dict(info, cam_infos={'CAM_FRONT': cam_q}), ['car'])
    c, s_ = np.cos(-0.5), np.sin(-0.5)
    want_xy = np.array([[c, -s_], [s_, c]]) @ np.array([30.0 - 1.0, -4.0 - 2.0])
    assert np.allclose(b2[0, :2].numpy(), want_xy, atol=1e-5) and abs(float(b2[0, 6]) - (yaw - 0.5)) < 1e-6
