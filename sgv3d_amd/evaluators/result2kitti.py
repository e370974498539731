"""Detections -> KITTI label files -> KITTI AP, as evaluators/result2kitti.py does it.

``result2kitti`` (:212-268): the results JSON written by ``RoadSideEvaluator._format_bbox`` (per image token a list of
``translation`` / ``size`` / ``box_yaw`` / ``detection_score`` / ``detection_name``) and the KITTI-format calibration
files of the data root (``training/calib/%06d.txt`` with ``P2:`` and ``Tr_velo_to_cam:`` rows, :200-210) -> one label
file per image under ``results_path/data``: class through ``category_map_dair`` (:16), detections above score 0.45,
``alpha`` from the box edge in the camera frame (:99-124), the 2-D box from the projected corners clipped to the image
(:157-173), location = the box's bottom centre in the camera frame, ``rotation_y = pi/2 - yaw``, every number rounded
to 4 decimals.  ``kitti_evaluation`` (:62-72): read the two label folders, run the KITTI evaluation (R40), write the
result text under ``metric_path/R40`` and return the moderate 3-D AP of ``Car``.  Host code (numpy), as in the
reference.  ``result2kitti_dair`` (:270-328) is the same conversion for the raw DAIR-V2X-I root (JSON calibration files,
float64) and ``result2kitti_rope3d`` (:330-393) for the raw Rope3D root: the lidar -> camera transform comes from the
image's ground-plane ("denorm") file (scripts/gen_info_rope3d.py:50-86), the camera matrix from the ``P2`` row of a calib
file named by the image token, and the output file name from a token -> sample-id JSON map."""
import json
import math
import os

import numpy as np

from .kitti_utils import kitti_common as kitti
from .kitti_utils.eval import kitti_eval

__all__ = ['kitti_evaluation', 'result2kitti', 'result2kitti_dair', 'result2kitti_rope3d', 'load_calib_dair',
           'load_calib_dair_json', 'load_calib_rope3d', 'category_map_dair', 'category_map_rope3d']


def kitti_evaluation(pred_label_path, gt_label_path, current_classes=("Car", "Pedestrian", "Cyclist"), metric_path="metric"):
    pred_annos, image_ids = kitti.get_label_annos(pred_label_path, return_ids=True)
    gt_annos = kitti.get_label_annos(gt_label_path, image_ids=image_ids)
    print(len(pred_annos), len(gt_annos))
    result, ret_dict = kitti_eval(gt_annos, pred_annos, current_classes=list(current_classes), metric="R40")
    mAP_3d_moderate = ret_dict["KITTI/Car_3D_moderate_strict"]
    os.makedirs(os.path.join(metric_path, "R40"), exist_ok=True)
    with open(os.path.join(metric_path, "R40", 'epoch_result_{}.txt'.format(round(mAP_3d_moderate, 2))), "w") as f:
        f.write(result)
    print(result)
    return mAP_3d_moderate


category_map_dair = {"car": "Car", "van": "Car", "truck": "Car", "bus": "Car", "pedestrian": "Pedestrian",
                     "bicycle": "Cyclist", "trailer": "Cyclist", "motorcycle": "Cyclist"}


category_map_rope3d = dict(category_map_dair)          # evaluators/result2kitti.py:17: the same mapping


def load_calib_dair(calib_file):
    """-> (Tr_velo_to_cam 4x4, camera matrix 3x3) from a KITTI-format calibration file; both rows are parsed in float32
    like the reference does (:200-210)."""
    P2 = Tr = None
    with open(calib_file, 'r') as f:
        for line in f:
            row = line.rstrip('\n').split(' ')
            if row[0] == 'P2:':
                P2 = np.array([float(v) for v in row[1:]], dtype=np.float32).reshape(3, 4)
            elif row[0] == 'Tr_velo_to_cam:':
                Tr = np.array([float(v) for v in row[1:]], dtype=np.float32).reshape(3, 4)
    Tr = np.concatenate((Tr, np.array([[0, 0, 0, 1]])), axis=0)
    return Tr, P2[:3, :3]


def _yaw_matrix(yaw):
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def _box_corners(size, yaw, bottom_centre):
    """8 corners [3, 8] of a box given (x extent, y extent, height), yaw about z and its bottom centre (:19-32,:99-110)."""
    l, w, h = size
    c = np.array([[l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2],
                  [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2],
                  [0, 0, 0, 0, h, h, h, h]], np.float64)
    return _yaw_matrix(yaw) @ c + np.asarray(bottom_centre, np.float64).reshape(3, 1)


def _normalize_angle(angle):
    a = np.arctan(np.tan(angle))                                    # :92-97
    return a + math.pi if np.cos(angle) < 0 else a


def load_calib_dair_json(dair_root, sample_id):
    """Calibration of the raw DAIR-V2X-I layout (:180-198): ``calib/camera_intrinsic/%06d.json`` (``cam_K``) and
    ``calib/virtuallidar_to_camera/%06d.json`` (``rotation`` + ``translation``, or ``Tr_velo_to_cam``), float64."""
    with open(os.path.join(dair_root, "calib/camera_intrinsic", "{:06d}".format(sample_id) + ".json")) as f:
        K = np.array(json.load(f)["cam_K"]).reshape([3, 3], order="C")
    with open(os.path.join(dair_root, "calib/virtuallidar_to_camera", "{:06d}".format(sample_id) + ".json")) as f:
        js = json.load(f)
    if "Tr_velo_to_cam" in js:
        v2c = np.array(js["Tr_velo_to_cam"]).reshape(3, 4)
        r, t = v2c[:, :3], v2c[:, 3].reshape(3, 1)
    else:
        r, t = np.array(js["rotation"]), np.array(js["translation"])
    Tr = np.eye(4)
    Tr[:3, :3] = r
    Tr[:3, 3] = t.flatten()
    return Tr, K


def load_calib_rope3d(rope_root, sample_token):
    """(Tr_velo_to_cam 4x4 float64, camera matrix 3x3 float32) of one Rope3D image: ``{training,validation}/denorm/
    <token>.txt`` holds the ground plane (a, b, c, d) in the camera frame, ``.../calib/<token>.txt`` the ``P2`` row
    (evaluators/result2kitti.py:337-345).  The camera -> lidar rotation turns the plane normal onto the camera's y axis
    (Rodrigues vector = unit axis in float32 times the angle, matrix rounded to float32 as the reference does), followed
    by the two axis permutations Rx, Rz; the lidar origin sits on the ground below the camera at distance |d| / |(a,b,c)|
    (scripts/gen_info_rope3d.py:56-86); Tr_velo_to_cam is its inverse (result2kitti.py:81-86)."""
    from ..input_contract import rodrigues
    sub = "training"
    if not os.path.exists(os.path.join(rope_root, "training/denorm", sample_token + ".txt")):
        sub = "validation"                                                          # :339-341
    with open(os.path.join(rope_root, sub, "denorm", sample_token + ".txt")) as f:
        denorm = np.array([float(v) for v in f.readlines()[0].split(' ')])
    P2 = None
    with open(os.path.join(rope_root, sub, "calib", sample_token + ".txt")) as f:
        for line in f:
            row = line.rstrip('\n').split(' ')
            if row[0] == 'P2:':
                P2 = np.array([float(v) for v in row[1:]], dtype=np.float32).reshape(3, 4)
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])
    Rz = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    origin = np.array([0, 1, 0])
    target = -1 * denorm[:3]
    target = target / np.sqrt(target[0] ** 2 + target[1] ** 2 + target[2] ** 2)
    sita = math.acos(np.inner(target, origin))
    n = np.cross(target, origin)
    n = (n / np.sqrt(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)).astype(np.float32)
    cam2lidar = rodrigues(n * sita).astype(np.float32)
    cam2lidar = Rz @ (Rx @ cam2lidar)
    Tr_cam2lidar = np.eye(4)
    Tr_cam2lidar[:3, :3] = cam2lidar
    Tr_cam2lidar[:3, 3] = [0, 0, abs(denorm[3]) / np.sqrt(np.sum(np.square(denorm[:3])))]
    return np.linalg.inv(Tr_cam2lidar), P2[:3, :3]


def _numeric_id(sample_token):
    return int(sample_token.split("/")[-1].split(".")[0])


def result2kitti(results_file, results_path, dair_root, gt_label_path, demo=False):
    """KITTI-layout data roots ('dair-v2x-i-kitti', 'rope3d-kitti'), :212-268."""
    return _convert(results_file, results_path, _numeric_id,
                    lambda tok: load_calib_dair(os.path.join(dair_root, "training/calib", "{:06d}".format(_numeric_id(tok)) + ".txt")),
                    category_map_dair)


def result2kitti_dair(results_file, results_path, dair_root, gt_label_path, demo=False):
    """Raw DAIR-V2X-I root (what exps/bevheight/dair-v2x/*.py configure: data_root 'data/dair-v2x-i/'), :270-328."""
    return _convert(results_file, results_path, _numeric_id, lambda tok: load_calib_dair_json(dair_root, _numeric_id(tok)),
                    category_map_dair)


def result2kitti_rope3d(results_file, results_path, dair_root, gt_label_path, demo=False,
                        token_map="data/rope3d-kitti/map_token2id.json"):
    """Raw Rope3D root (exps/bevheight/rope3d/*.py), :330-393.  ``token_map`` is the JSON the reference opens at this
    fixed relative path (:333-334): image token -> sample id of the KITTI-format copy, which names the label file."""
    with open(token_map) as fp:
        token2sample = json.load(fp)
    return _convert(results_file, results_path, lambda tok: int(token2sample[tok]),
                    lambda tok: load_calib_rope3d(dair_root, tok), category_map_rope3d)


def _convert(results_file, results_path, sample_id_of, load_calib, category_map):
    with open(results_file, 'r', encoding='utf8') as fp:
        results = json.load(fp)["results"]
    os.makedirs(os.path.join(results_path, "data"), exist_ok=True)
    for sample_token, preds in results.items():
        sample_id = sample_id_of(sample_token)
        Tr, K = load_calib(sample_token)
        R, t = Tr[:3, :3].astype(np.float64), Tr[:3, 3].astype(np.float64).reshape(3, 1)
        K34 = np.concatenate([K, np.zeros((3, 1))], axis=1)
        lines = []
        for pred in preds:
            x, y, z = pred["translation"]
            w, l, h = pred["size"]
            yaw_lidar = pred["box_yaw"]
            score, cls = pred["detection_score"], pred["detection_name"]
            centre_cam = R @ np.array([[x], [y], [z]], np.float64) + t
            # orientation: direction of the box edge corner 3 -> corner 0 in the camera's x-z plane (:112-124)
            cam = R @ _box_corners([l, w, h], yaw_lidar, [x, y, z]) + t
            yaw_cam = math.atan2(-(cam[2, 0] - cam[2, 3]), cam[0, 0] - cam[0, 3])
            alpha = yaw_cam - math.atan2(centre_cam[0, 0], centre_cam[2, 0])
            if alpha > math.pi:
                alpha -= 2.0 * math.pi
            if alpha <= -math.pi:
                alpha += 2.0 * math.pi
            alpha = _normalize_angle(alpha)
            rot_y = 0.5 * np.pi - yaw_lidar                                                   # :239
            cam_xyz = (Tr @ np.array([x, y, z, 1]))[:3]
            # 2-D box: the reference builds these corners with (w, l, h) as extents and the centre raised by h/2, whose
            # bottom is lowered again inside get_lidar_3d_8points (:241-243, :19-32)
            corners = _box_corners([w, l, h], yaw_lidar, [x, y, (z + h / 2) - h / 2]).T       # [8, 3]
            hom = Tr @ np.concatenate([corners, np.ones((8, 1), dtype=np.float32)], axis=1).T
            uv = K34 @ hom
            uv = uv[:2] / uv[2]
            box2d = np.array([max(uv[0].min(), 0.0), max(uv[1].min(), 0.0), min(uv[0].max(), 1920.0), min(uv[1].max(), 1080.0)])
            if score > 0.45 and cls in category_map:
                r4 = lambda v: str(round(v, 4))
                lines.append([category_map[cls], "0", "0", r4(alpha)] + [r4(v) for v in box2d] +
                             [r4(h), r4(l), r4(w), r4(cam_xyz[0]), r4(cam_xyz[1]), r4(cam_xyz[2]), r4(rot_y), r4(score)])
        with open(os.path.join(results_path, "data", "{:06d}".format(sample_id) + ".txt"), "w") as f:
            for line in lines:
                f.write(" ".join(line) + "\n")
    return os.path.join(results_path, "data")
