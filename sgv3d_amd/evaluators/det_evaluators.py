"""``RoadSideEvaluator`` of evaluators/det_evaluators.py:17-176: detections of ``BEVHeight.get_bboxes`` (per image
``(boxes [n, 9], scores, labels)``) + the images' meta dicts -> results JSON -> KITTI label files -> KITTI AP.

The reference builds every box through nuscenes-devkit's ``Box`` and pyquaternion (``rotate`` by ego2global, then
``translate``, :121-130); neither package is a dependency here, the two operations are written out (unit quaternion ->
rotation matrix; the yaw quaternion composed with the ego2global one).  For the roadside data sets ego2global is the
identity.  PARITY UNPINNED for this class (the two packages are absent from the build image); ``result2kitti`` and
``kitti_evaluation`` behind it are pinned by the reference's own outputs."""
import json
import os
import os.path as osp
import tempfile

import numpy as np

from .result2kitti import kitti_evaluation, result2kitti, result2kitti_dair, result2kitti_rope3d

__all__ = ['RoadSideEvaluator']


def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _quat_matrix(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class RoadSideEvaluator():
    DefaultAttribute = {
        'car': 'vehicle.parked', 'pedestrian': 'pedestrian.moving', 'trailer': 'vehicle.parked', 'truck': 'vehicle.parked',
        'bus': 'vehicle.moving', 'motorcycle': 'cycle.without_rider', 'construction_vehicle': 'vehicle.parked',
        'bicycle': 'cycle.without_rider', 'barrier': '', 'traffic_cone': '',
    }

    def __init__(self, class_names, current_classes, data_root, gt_label_path,
                 modality=dict(use_lidar=False, use_camera=True, use_radar=False, use_map=False, use_external=False),
                 output_dir=None):
        self.class_names = class_names
        self.current_classes = current_classes
        self.data_root = data_root
        self.gt_label_path = gt_label_path
        self.modality = modality
        self.output_dir = output_dir

    def format_results(self, results, img_metas, result_names=['img_bbox'], jsonfile_prefix=None, **kwargs):
        assert isinstance(results, list), 'results must be a list'
        if jsonfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = osp.join(tmp_dir.name, 'results')
        else:
            tmp_dir = None
        result_files = dict()
        for name in result_names:
            if '2d' in name:
                continue
            target = self.output_dir if self.output_dir else osp.join(jsonfile_prefix, name)
            result_files[name] = self._format_bbox(results, img_metas, target)
        return result_files, tmp_dir

    def evaluate(self, results, img_metas, metric='bbox', logger=None, jsonfile_prefix=None, result_names=['img_bbox'],
                 show=False, out_dir=None, pipeline=None, results_path="outputs", metric_path="outputs/metrics"):
        """:83-107: KITTI-layout roots, the raw DAIR-V2X-I root (the one the dair-v2x experiment files configure) and,
        for any other root, the raw Rope3D layout (exps/bevheight/rope3d/*.py)."""
        result_files, tmp_dir = self.format_results(results, img_metas, result_names, jsonfile_prefix)
        if 'dair-v2x-i-kitti' in self.data_root or 'rope3d-kitti' in self.data_root:
            convert = result2kitti
        elif 'dair-v2x-i' in self.data_root:
            convert = result2kitti_dair
        else:
            convert = result2kitti_rope3d
        pred_label_path = convert(result_files["img_bbox"], results_path, self.data_root, self.gt_label_path, demo=False)
        return kitti_evaluation(pred_label_path, self.gt_label_path, current_classes=self.current_classes,
                                metric_path=metric_path)

    def _format_bbox(self, results, img_metas, jsonfile_prefix=None):
        annos_by_token = {}
        for sample_id, det in enumerate(results):
            boxes, scores, labels = det
            boxes = np.asarray(boxes.tensor.cpu() if hasattr(boxes, 'tensor') else
                               (boxes.cpu() if hasattr(boxes, 'cpu') else boxes), np.float64).reshape(-1, 9)
            scores = np.asarray(scores.cpu() if hasattr(scores, 'cpu') else scores, np.float64)
            labels = np.asarray(labels.cpu() if hasattr(labels, 'cpu') else labels).astype(np.int64)
            token = img_metas[sample_id]['token']
            trans = np.array(img_metas[sample_id]['ego2global_translation'], np.float64)
            rot = np.array(img_metas[sample_id]['ego2global_rotation'], np.float64)          # (w, x, y, z)
            Rm = _quat_matrix(rot)
            annos = []
            for i, box in enumerate(boxes):
                name = self.class_names[labels[i]]
                yaw = float(box[6])
                quat = _quat_mul(rot / np.linalg.norm(rot), np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]))
                centre = Rm @ box[:3] + trans
                vel = Rm @ np.array([box[7], box[8], 0.0])
                moving = np.sqrt(vel[0] ** 2 + vel[1] ** 2) > 0.2
                if moving and name in ('car', 'construction_vehicle', 'bus', 'truck', 'trailer'):
                    attr = 'vehicle.moving'
                elif moving and name in ('bicycle', 'motorcycle'):
                    attr = 'cycle.with_rider'
                elif not moving and name == 'pedestrian':
                    attr = 'pedestrian.standing'
                elif not moving and name == 'bus':
                    attr = 'vehicle.stopped'
                else:
                    attr = self.DefaultAttribute[name]
                annos.append(dict(sample_token=token, translation=centre.tolist(), size=box[[4, 3, 5]].tolist(),
                                  rotation=quat.tolist(), box_yaw=yaw, velocity=vel[:2].tolist(), detection_name=name,
                                  detection_score=float(scores[i]), attribute_name=attr))
            annos_by_token.setdefault(token, []).extend(annos)
        os.makedirs(jsonfile_prefix, exist_ok=True)
        res_path = osp.join(jsonfile_prefix, 'results_nusc.json')
        with open(res_path, 'w') as f:
            json.dump({'meta': self.modality, 'results': annos_by_token}, f)
        return res_path
