"""GPU: closed-loop check of the "mAP within 0.1 of the reference" target with what this image can offer (no data
set, no checkpoint): a synthetic scene set goes through BOTH chains

    HIP chain     images -> BEVHeight.forward (HIP kernels) -> get_bboxes (device decode + circle NMS)
    oracle chain  images -> oracle/torch_model.py (torch-CPU fp32) -> oracle/decode_ref.py

and then through the same evaluator code (RoadSideEvaluator -> results JSON -> result2kitti label files -> KITTI AP
R40, the reference's evaluation path, evaluators/det_evaluators.py:83-106), against ground truth derived from the
oracle's own detections (kept, jittered or dropped boxes), so that the AP values are neither 0 nor 100.  With the kernels
whose f32 rounding is closest to the oracle's (fixed per-layer rule, F(2x2) fused head: ~1e-6 from the oracle) the two result
texts must be identical character by character: no detection, no match and no AP digit changes.  With the kernels as shipped
(F(4x4) fused head: ~1e-5) every AP value must stay within 0.02 of the oracle chain's -- a fifth of the 0.1 the target allows.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref, torch_model as TM
from sgv3d_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_FRAMES = 12


def _calib_text(c):
    """KITTI calib file (P2 + Tr_velo_to_cam rows, the layout evaluators/result2kitti.py:200-210 reads) of a
    synthetic camera: lidar/ego frame = the model's ego frame, camera matrix at the full 1080x1920 resolution."""
    P2 = np.zeros((3, 4))
    P2[:3, :3] = c['intrin'][:3, :3]
    tr = np.linalg.inv(c['sensor2ego'].astype(np.float64))[:3]
    return "P2: " + " ".join(f"{v:.6f}" for v in P2.reshape(-1)) + "\n" + \
           "Tr_velo_to_cam: " + " ".join(f"{v:.8f}" for v in tr.reshape(-1)) + "\n"


@pytest.fixture
def fixed_kernel_choice():
    """The per-layer first-call measurement picks among kernels whose f32 sums differ in association (split-K, Winograd): with it
    the HIP outputs move by ~1e-6 from run to run, which can flip one near-threshold 2-D box match and with it a fourth AP
    digit.  The fixed rule makes the HIP chain the same computation on every run."""
    from sgv3d_amd import hip_ops
    old = hip_ops.AUTOTUNE
    hip_ops.AUTOTUNE = False
    yield
    hip_ops.AUTOTUNE = old


def test_hip_and_oracle_chains_give_identical_ap_text(tmp_path, fixed_kernel_choice):
    from sgv3d_amd.evaluators import RoadSideEvaluator
    from sgv3d_amd.evaluators.kitti_utils import eval as E, kitti_common as KC
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.small_conf(final=(128, 192), bev=64, depth=18)
    # head geometry of the reduced grid: 64 x 64 cells of 0.4 m in front of the camera
    hc['bbox_coder'] = dict(hc['bbox_coder'], pc_range=[0, -12.8, -5, 25.6, 12.8, 3], post_center_range=[0.0, -15.0, -10.0, 30.0, 15.0, 10.0],
                            max_num=100)
    hc['test_cfg'] = dict(hc['test_cfg'], post_center_limit_range=[0.0, -15.0, -10.0, 30.0, 15.0, 10.0], post_max_size=20)
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.checkpoint_like_(m, 2)          # trained-checkpoint-like statistics (the weights of tests/test_fullsize_gpu.py's extra parity test)
    with torch.no_grad():          # an untrained heatmap sits at sigmoid(-2.19): spread the logits so that some cells fire
        for t in m.head.task_heads:
            t.heatmap[1].weight.mul_(40.0)
            t.heatmap[1].bias.fill_(-1.0)
            t.dim[1].weight.mul_(3.0)
            t.dim[1].bias.fill_(0.6)          # exp() -> sizes around 1.8 m
            t.height[1].bias.fill_(-0.5)
    imgs = torch.cat([S.make_images(1, bc['final_dim'], seed=100 + i) for i in range(N_FRAMES)])
    cams = [S.make_calib(pitch_deg=11.0 + (i % 3), cam_h=5.5 + 0.3 * (i % 2), fx=2183.375 * 192 / 1536, fy=2329.2976 * 128 / 864,
                         cx=940.59 * 192 / 1536, cy=567.568 * 128 / 864) for i in range(N_FRAMES)]
    t = lambda k: torch.from_numpy(np.stack([c[k] for c in cams])).view(N_FRAMES, 1, 1, 4, 4)
    mats = {'sensor2ego_mats': t('sensor2ego'), 'intrin_mats': t('intrin'), 'ida_mats': t('ida'),
            'sensor2sensor_mats': torch.eye(4).view(1, 1, 1, 4, 4).repeat(N_FRAMES, 1, 1, 1, 1),
            'sensor2virtual_mats': t('sensor2virtual'),
            'reference_heights': torch.tensor([float(c['reference_height']) for c in cams]).view(N_FRAMES, 1, 1),
            'bda_mat': torch.eye(4).repeat(N_FRAMES, 1, 1)}
    # ---- oracle chain ---------------------------------------------------------------------------------------------
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref = TM.bevheight_forward(sd, bc, hc, imgs, mats)
    ref_np = tuple([{k: v.numpy() for k, v in task[0].items()}] for task in ref)
    nc = [len(tk['class_names']) for tk in hc['tasks']]
    det_oracle = [(b, s, l) for b, s, l in decode_ref.get_bboxes(ref_np, hc['bbox_coder'], hc['test_cfg'], nc)]
    # ---- HIP chain, twice: F(2x2) fused head (closest rounding) and the shipped default (F(4x4) fused head) ----------------
    from sgv3d_amd import hip_ops
    m = m.to(DEV)
    det_hips = {}
    old_path = hip_ops.HEAD_PATH
    try:
        for tag, path in (('hip', 0), ('hip_default', old_path)):
            hip_ops.HEAD_PATH = path
            with torch.no_grad():
                preds = m(imgs.to(DEV), {k: v.to(DEV) for k, v in mats.items()})
                out = m.get_bboxes(preds)
            det_hips[tag] = [(b.tensor.cpu().numpy(), s.cpu().numpy(), l.cpu().numpy()) for b, s, l in out]
    finally:
        hip_ops.HEAD_PATH = old_path
    n_det = sum(len(s) for _, s, _ in det_oracle)
    assert n_det > 10 * N_FRAMES, n_det                       # a scene set with plenty of detections
    for tag, det_hip in det_hips.items():
        for (bo, so, lo), (bh, sh, lh) in zip(det_oracle, det_hip):
            assert len(so) == len(sh) and np.array_equal(lo, lh), tag
            np.testing.assert_allclose(bh, bo, rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(sh, so, rtol=1e-5, atol=1e-5)
    # ---- data root + ground truth derived from the oracle's detections ------------------------------------------------
    root = tmp_path / 'dair-v2x-i-kitti'
    os.makedirs(root / 'training' / 'calib')
    metas = []
    for i, c in enumerate(cams):
        # image-plane quantities at the resolution the KITTI tools assume (1920x1080): undo the intrinsics shrink
        full = dict(c, intrin=np.diag([1536 / 192, 864 / 128, 1, 1]).astype(np.float32) @ c['intrin'] / 0.8)
        full['intrin'][2, 2] = 1.0
        (root / 'training' / 'calib' / f'{i:06d}.txt').write_text(_calib_text(full))
        metas.append(dict(token=f'training/image_2/{i:06d}.jpg', ego2global_translation=[0, 0, 0], ego2global_rotation=[1, 0, 0, 0]))
    names = S.CLASSES
    rng = np.random.default_rng(5)
    gt_dets = []
    for b, s, l in det_oracle:
        keep = (s > 0.45) & (rng.uniform(size=len(s)) < 0.8)             # drop a fifth: false positives for both chains
        gb = b[keep].copy()
        gb[:, :2] += rng.normal(0, 0.15, (len(gb), 2)).astype(np.float32)   # jitter: IoU < 1, some matches fail at 0.7 / 0.5
        gb[:, 6] += rng.normal(0, 0.05, len(gb)).astype(np.float32)
        gt_dets.append((gb, np.ones(len(gb), np.float32), l[keep]))

    def labels_of(dets, tag):
        ev = RoadSideEvaluator(class_names=names, current_classes=["Car", "Pedestrian", "Cyclist"], data_root=str(root),
                               gt_label_path=str(tmp_path / 'gt'))
        files, _ = ev.format_results(dets, metas, jsonfile_prefix=str(tmp_path / f'json_{tag}'))
        from sgv3d_amd.evaluators.result2kitti import result2kitti
        return result2kitti(files['img_bbox'], str(tmp_path / f'out_{tag}'), str(root), str(tmp_path / 'gt'))
    gt_raw = labels_of(gt_dets, 'gt')
    os.makedirs(tmp_path / 'gt')
    for i in range(N_FRAMES):                                               # ground-truth files: no score column
        lines = open(os.path.join(gt_raw, f'{i:06d}.txt')).read().splitlines()
        (tmp_path / 'gt' / f'{i:06d}.txt').write_text("".join(" ".join(ln.split(' ')[:15]) + "\n" for ln in lines))
    texts, aps = {}, {}
    for tag, dets in (('oracle', det_oracle), ('hip', det_hips['hip']), ('hip_default', det_hips['hip_default'])):
        path = labels_of(dets, tag)
        dt, ids = KC.get_label_annos(path, return_ids=True)
        gt = KC.get_label_annos(str(tmp_path / 'gt'), image_ids=ids)
        texts[tag], ret = E.kitti_eval(gt, dt, ["Car", "Pedestrian", "Cyclist"], metric="R40")
        aps[tag] = ret
        if tag == 'oracle':
            vals = np.array(list(ret.values()))
            assert (vals > 1).any() and (vals < 99).any(), vals          # a non-degenerate AP table
    assert texts['hip'] == texts['oracle']
    assert aps['hip_default'].keys() == aps['oracle'].keys()
    worst = max(abs(float(aps['hip_default'][k]) - float(aps['oracle'][k])) for k in aps['oracle'])
    print(f"closed loop: {n_det} detections in {N_FRAMES} frames; shipped kernels: largest |AP - oracle chain's AP| = {worst:.4f}, "
          f"text identical: {texts['hip_default'] == texts['oracle']}")
    assert worst <= 0.02, (worst, texts['hip_default'], texts['oracle'])
