"""KITTI-AP evaluator on the MI355X: the rotated-box kernel against the reference's own overlap
matrices and the float64 oracle, and ``kitti_eval`` / ``kitti_evaluation`` end to end against the reference's result
text and values (tests/golden/kitti_eval.npz)."""
import os

import numpy as np
import pytest

from oracle import kitti_eval_ref as R
from sgv3d_amd.evaluators import kitti_evaluation
from sgv3d_amd.evaluators.kitti_utils import eval as E
from sgv3d_amd.evaluators.kitti_utils import kitti_common as KC
from sgv3d_amd.evaluators.kitti_utils.rotate_iou import rotate_iou_gpu_eval, rotate_iou_pairs

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "kitti_eval.npz"))


def _write(tmp_path):
    for sub, texts in (('gt', GOLD['label_gt']), ('dt', GOLD['label_dt'])):
        os.makedirs(tmp_path / sub)
        for i, t in enumerate(texts):
            (tmp_path / sub / f'{i:06d}.txt').write_text(str(t))
    return str(tmp_path / 'dt'), str(tmp_path / 'gt')


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotated_overlap_kernel_matches_reference(criterion):
    b, q = GOLD['riou_boxes'], GOLD['riou_qboxes']
    got = rotate_iou_gpu_eval(b, q, criterion)
    want = GOLD[f'riou_c{criterion}']
    assert got.shape == want.shape and got.dtype == b.dtype
    # float32 in the reference's operation order, touching / identical boxes included (same inside / crossing decisions)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6 * float(np.abs(want).max()))
    assert ((got > 0) == (want > 0)).all()


def test_3d_overlap_kernel_matches_reference_and_oracle():
    b, q = GOLD['d3_boxes'], GOLD['d3_qboxes']
    got = E.d3_box_overlap(b, q)
    np.testing.assert_allclose(got, GOLD['d3_overlap'], rtol=2e-6, atol=2e-7)
    for c in (0, 1):
        np.testing.assert_allclose(rotate_iou_pairs([b], [q], c)[0], R.d3_overlap(b, q, c), rtol=0, atol=2e-5)


def test_ragged_launch_equals_per_image_calls_and_oracle():
    rng = np.random.default_rng(3)
    mk = lambda n: np.stack([rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(0.5, 6, n), rng.uniform(0.5, 3, n),
                             rng.uniform(-4, 4, n)], 1).reshape(n, 5)
    ns, ks = [0, 1, 17, 40, 5, 16, 33], [3, 0, 16, 9, 70, 1, 33]
    bl, ql = [mk(n) for n in ns], [mk(k) for k in ks]
    outs = rotate_iou_pairs(bl, ql, -1)
    for b, q, o in zip(bl, ql, outs):
        assert o.shape == (len(b), len(q))
        if len(b) and len(q):
            assert np.array_equal(o, rotate_iou_pairs([b], [q], -1)[0])          # tile decomposition does not matter
            assert np.abs(o - R.rotated_overlap(b, q)).max() <= 2e-5


def test_kitti_eval_matches_reference_text_and_values(tmp_path):
    dt_dir, gt_dir = _write(tmp_path)
    dt_annos, ids = KC.get_label_annos(dt_dir, return_ids=True)
    gt_annos = KC.get_label_annos(gt_dir, image_ids=ids)
    result, ret = E.kitti_eval(gt_annos, dt_annos, ["Car", "Pedestrian", "Cyclist"], metric="R40")
    keys = [str(k) for k in GOLD['ret_keys']]
    assert sorted(ret) == keys
    got = np.array([ret[k] for k in keys])
    np.testing.assert_allclose(got, GOLD['ret_vals'], rtol=0, atol=1e-9)
    assert result == str(GOLD['result_text'])
    _, ret11 = E.kitti_eval(gt_annos, dt_annos, ["Car", "Pedestrian", "Cyclist"], eval_types=['bbox', 'bev', '3d'], metric="R11")
    np.testing.assert_allclose(np.array([ret11[k] for k in keys]), GOLD['ret11_vals'], rtol=0, atol=1e-9)
    assert float(GOLD['ret_vals'].max()) > 30 and (GOLD['ret_vals'] > 0).mean() > 0.7        # a fixture with real matches


def test_kitti_evaluation_writes_the_result_file(tmp_path, capsys):
    dt_dir, gt_dir = _write(tmp_path)
    ap = kitti_evaluation(dt_dir, gt_dir, metric_path=str(tmp_path / 'metrics'))
    keys = [str(k) for k in GOLD['ret_keys']]
    want = float(GOLD['ret_vals'][keys.index('KITTI/Car_3D_moderate_strict')])
    assert abs(ap - want) < 1e-9
    files = os.listdir(tmp_path / 'metrics' / 'R40')
    assert files == ['epoch_result_{}.txt'.format(round(want, 2))]
    assert (tmp_path / 'metrics' / 'R40' / files[0]).read_text() == str(GOLD['result_text'])


def test_roadside_evaluator_end_to_end(tmp_path):
    """get_bboxes-style detections -> RoadSideEvaluator.evaluate -> JSON -> label files -> KITTI AP, with the label set the
    reference's result2kitti wrote for the same detections as ground truth."""
    import json
    from sgv3d_amd.evaluators import RoadSideEvaluator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "result2kitti.npz"))
    root = tmp_path / 'dair-v2x-i-kitti'
    os.makedirs(root / 'training' / 'calib')
    os.makedirs(tmp_path / 'gt')
    for sid, calib, lab in zip(g['calib_ids'], g['calib_text'], g['label_text']):
        (root / 'training' / 'calib' / f'{int(sid):06d}.txt').write_text(str(calib))
        (tmp_path / 'gt' / f'{int(sid):06d}.txt').write_text("\n".join(" ".join(ln.split(' ')[:15]) for ln in str(lab).splitlines()) + "\n")
    names = ['car', 'van', 'truck', 'bus', 'pedestrian', 'bicycle', 'trailer', 'motorcycle', 'barrier']
    results, metas = [], []
    for token, preds in json.loads(str(g['results_json']))['results'].items():
        boxes = np.array([p['translation'] + [p['size'][1], p['size'][0], p['size'][2], p['box_yaw'], 0.0, 0.0] for p in preds])
        results.append((boxes, np.array([p['detection_score'] for p in preds]), np.array([names.index(p['detection_name']) for p in preds])))
        metas.append(dict(token=token, ego2global_translation=[0, 0, 0], ego2global_rotation=[1, 0, 0, 0]))
    ev = RoadSideEvaluator(class_names=names, current_classes=["Car", "Pedestrian", "Cyclist"], data_root=str(root),
                           gt_label_path=str(tmp_path / 'gt'))
    ap = ev.evaluate(results, metas, jsonfile_prefix=str(tmp_path / 'json'), results_path=str(tmp_path / 'outputs'),
                     metric_path=str(tmp_path / 'metrics'))
    for sid, lab in zip(g['calib_ids'], g['label_text']):
        assert (tmp_path / 'outputs' / 'data' / f'{int(sid):06d}.txt').read_text() == str(lab)
    # (identical boxes are the degenerate case of the reference's float32 clipping -- every corner on the other outline --
    # and three images give the 40-point AP only a few recall steps: the value itself says little here; the chain does)
    assert 0.0 <= ap <= 100.0
    files = os.listdir(tmp_path / 'metrics' / 'R40')
    text = (tmp_path / 'metrics' / 'R40' / files[0]).read_text()
    assert 'Car AP@0.70, 0.70, 0.70:' in text and 'Overall AP@easy, moderate, hard:' in text
    assert float(text.split('bbox AP:')[1].split(',')[0]) > 0
