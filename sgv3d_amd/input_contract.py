"""The producer side of ``mats_dict`` (first half): the geometry helpers the
reference's dataset runs per camera before the forward (dataset/nusc_mv_det_dataset.py:41-86, 433-454,
783-787, 855-871), restated so that a caller without the reference's data pipeline (mmcv / nuscenes /
cv2 are not dependencies here) can build the exact tensors ``BEVHeight.forward`` expects.

Pinned by ``tests/golden/input_contract.npz``: outputs of the reference's own ``equation_plane`` /
``get_denorm`` / ``get_sensor2virtual`` / ``get_reference_height`` / ``sample_ida_augmentation`` executed
in the build container (``tests/golden/make_golden.py``; cv2.Rodrigues replaced by the closed-form
rotation formula it implements).  Host-side numpy on 4x4 matrices; nothing here is on the hot path.
"""
import math

import numpy as np
import torch

__all__ = ['equation_plane', 'get_denorm', 'get_sensor2virtual', 'get_reference_height', 'rodrigues',
           'ida_resize_crop', 'ida_matrix', 'bda_matrix', 'collate_mats', 'camera_from_info', 'mats_from_infos',
           'gt_from_info', 'NAME_TO_DETECTION_CLASS']


def rodrigues(rvec):
    """Rotation matrix of the axis-angle vector ``rvec`` (what cv2.Rodrigues returns for a 3-vector)."""
    rvec = np.asarray(rvec, dtype=np.float64)
    th = float(np.linalg.norm(rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * kx + (1 - math.cos(th)) * (kx @ kx)


def equation_plane(points):
    """Plane a x + b y + c z + d = 0 through three points (dataset/...:41-61)."""
    (x1, y1, z1), (x2, y2, z2), (x3, y3, z3) = (p[:3] for p in points)
    a1, b1, c1 = x2 - x1, y2 - y1, z2 - z1
    a2, b2, c2 = x3 - x1, y3 - y1, z3 - z1
    a = b1 * c2 - b2 * c1
    b = a2 * c1 - a1 * c2
    c = a1 * b2 - b1 * a2
    return np.array([a, b, c, -a * x1 - b * y1 - c * z1])


def get_denorm(sweepego2sweepsensor):
    """Ground plane (ego z = 0) in camera coordinates, sign-flipped (dataset/...:63-68)."""
    gp = np.array([[0.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 1.0, 0.0, 1.0]])
    cam = np.matmul(np.asarray(sweepego2sweepsensor), gp.T).T
    return -1 * equation_plane(cam)


def get_sensor2virtual(denorm):
    """Rotation that turns the camera so that its y axis is the ground normal (dataset/...:70-82)."""
    origin = np.array([0, 1, 0])
    target = -1 * np.array([denorm[0], denorm[1], denorm[2]])
    target = target / np.sqrt(target[0] ** 2 + target[1] ** 2 + target[2] ** 2)
    sita = math.acos(np.inner(target, origin))
    n = np.cross(target, origin)
    n = n / np.sqrt(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)
    n = n.astype(np.float32)
    rot = rodrigues(n * sita).astype(np.float32)
    out = np.eye(4)
    out[:3, :3] = rot
    return out.astype(np.float32)


def get_reference_height(denorm):
    """Camera height above the ground plane (dataset/...:84-86)."""
    return (np.abs(denorm[3]) / np.sqrt(denorm[0] ** 2 + denorm[1] ** 2 + denorm[2] ** 2)).astype(np.float32)


def ida_resize_crop(src_hw, final_dim, bot_pct_lim=(0.0, 0.0)):
    """``sample_ida_augmentation`` in eval mode (dataset/...:433-446): (resize, resize_dims, crop, flip, rotate)."""
    H, W = src_hw
    fH, fW = final_dim
    resize = max(fH / H, fW / W)
    newW, newH = int(W * resize), int(H * resize)
    crop_h = int((1 - np.mean(bot_pct_lim)) * newH) - fH
    crop_w = int(max(0, newW - fW) / 2)
    return resize, (newW, newH), (crop_w, crop_h, crop_w + fW, crop_h + fH), False, 0


def ida_matrix(resize, crop, flip=False, rotate_deg=0.0):
    """4x4 image-data-augmentation matrix of ``img_transform`` (dataset/...:88-122) for the eval settings
    (resize + crop; flip / rotate about the crop centre supported for completeness)."""
    rot = np.eye(2) * resize
    tran = -np.array(crop[:2], dtype=np.float64)
    if flip:
        a = np.array([[-1.0, 0.0], [0.0, 1.0]])
        b = np.array([crop[2] - crop[0], 0.0])
        rot, tran = a @ rot, a @ tran + b
    th = rotate_deg / 180.0 * np.pi
    a = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
    b = np.array([crop[2] - crop[0], crop[3] - crop[1]]) / 2.0
    b = a @ (-b) + b
    rot, tran = a @ rot, a @ tran + b
    out = np.eye(4)
    out[:2, :2] = rot
    out[:2, 3] = tran
    return out.astype(np.float32)


def bda_matrix(rotate_deg=0.0, scale=1.0, flip_dx=False, flip_dy=False):
    """4x4 BEV-data-augmentation matrix (dataset/...:124-146, 783-787); identity for evaluation."""
    th = rotate_deg / 180.0 * np.pi
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]) * scale
    if flip_dx:
        rot = np.diag([-1.0, 1.0, 1.0]) @ rot
    if flip_dy:
        rot = np.diag([1.0, -1.0, 1.0]) @ rot
    out = np.eye(4)
    out[:3, :3] = rot
    return out.astype(np.float32)


def collate_mats(cameras, device='cpu'):
    """``collate_fn`` layout (dataset/...:855-871) for one sweep / one camera per sample.  ``cameras``:
    list of dicts with 4x4 'sensor2ego', 'intrin', 'ida', 'bda' (optional 'sensor2sensor'); the
    sensor2virtual matrix and the reference height are derived here as the dataset does."""
    s2e, K, ida, s2s, s2v, bda, refh = [], [], [], [], [], [], []
    for c in cameras:
        e2s = np.linalg.inv(np.asarray(c['sensor2ego'], dtype=np.float64))
        denorm = get_denorm(e2s)
        s2e.append(np.asarray(c['sensor2ego'], np.float32))
        K.append(np.asarray(c['intrin'], np.float32))
        ida.append(np.asarray(c['ida'], np.float32))
        s2s.append(np.asarray(c.get('sensor2sensor', np.eye(4)), np.float32))
        s2v.append(get_sensor2virtual(denorm))
        refh.append(get_reference_height(denorm))
        bda.append(np.asarray(c.get('bda', np.eye(4)), np.float32))
    n = len(cameras)
    t = lambda xs: torch.from_numpy(np.stack(xs)).view(n, 1, 1, 4, 4).to(device)
    return {'sensor2ego_mats': t(s2e), 'intrin_mats': t(K), 'ida_mats': t(ida), 'sensor2sensor_mats': t(s2s),
            'sensor2virtual_mats': t(s2v),
            'reference_heights': torch.from_numpy(np.asarray(refh, np.float32)).view(n, 1, 1).to(device),
            'bda_mat': torch.from_numpy(np.stack(bda)).to(device)}


# ------------------------------------------------------------------------------------------------ info records (.pkl schema)
# category_name -> detection class, dataset/nusc_mv_det_dataset.py:18-39 (the roadside infos of scripts/gen_info_dair.py
# already carry detection names, which map to themselves)
NAME_TO_DETECTION_CLASS = {n: n for n in ('car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle',
                                          'bicycle', 'pedestrian', 'traffic_cone')}


def _quat_to_matrix(q):
    w, x, y, z = (float(v) for v in q)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def camera_from_info(cam_info, final_dim, bot_pct_lim=(0.0, 0.0)):
    """One camera record of an info entry (scripts/gen_info_dair.py:125-147: 'calibrated_sensor' with 'rotation_matrix' or
    a 'rotation' quaternion, 'translation', 'camera_intrinsic'; 'height' / 'width') -> the dict ``collate_mats`` takes, with
    the evaluation-time image transform (resize to ``final_dim``, no flip / rotation; dataset/...:433-446, 520-531)."""
    cs = cam_info['calibrated_sensor']
    rot = np.asarray(cs['rotation_matrix'], np.float64) if 'rotation_matrix' in cs else _quat_to_matrix(cs['rotation'])
    s2e = np.eye(4)
    s2e[:3, :3] = rot
    s2e[:3, 3] = np.asarray(cs['translation'], np.float64).reshape(3)
    K = np.eye(4)
    K[:3, :3] = np.asarray(cs['camera_intrinsic'], np.float64).reshape(3, 3)
    resize, _, crop, flip, rotate = ida_resize_crop((int(cam_info.get('height', 1080)), int(cam_info.get('width', 1920))),
                                                    final_dim, bot_pct_lim)
    return dict(sensor2ego=s2e, intrin=K, ida=ida_matrix(resize, crop, flip, rotate), bda=np.eye(4))


def mats_from_infos(infos, final_dim, cam='CAM_FRONT', device='cpu'):
    """``mats_dict`` of a batch of info entries (one key frame, one camera each: every shipped roadside config)."""
    return collate_mats([camera_from_info(info['cam_infos'][cam], final_dim) for info in infos], device)


def gt_from_info(info, classes, cam='CAM_FRONT'):
    """``get_gt`` (dataset/...:668-712): annotations -> (boxes float32 [n, 9] = x, y, z, dx (l), dy (w), dz (h), yaw, vx, vy
    in the ego frame; labels int64 [n]).  'size' is (w, l, h) as nuscenes' Box keeps it; the yaw is the rotation of the
    box's quaternion (or 'yaw_lidar') composed with the inverse ego pose."""
    ego = info['cam_infos'][cam]['ego_pose']
    R = _quat_to_matrix(ego['rotation']).T                       # global -> ego
    t = -np.asarray(ego['translation'], np.float64)
    boxes, labels = [], []
    for ann in info['ann_infos']:
        name = NAME_TO_DETECTION_CLASS.get(ann['category_name'])
        if name not in classes or ann['num_lidar_pts'] + ann['num_radar_pts'] <= 0:
            continue
        centre = R @ (np.asarray(ann['translation'], np.float64) + t)
        q = ann['rotation']
        q = getattr(q, 'elements', q)                            # pyquaternion object or a plain (w, x, y, z)
        Rb = R @ _quat_to_matrix(q)
        yaw = math.atan2(Rb[1, 0], Rb[0, 0])
        w, l, h = (float(v) for v in ann['size'])
        vel = R @ np.asarray(ann.get('velocity', np.zeros(3)), np.float64).reshape(3)
        boxes.append([centre[0], centre[1], centre[2], l, w, h, yaw, vel[0], vel[1]])
        labels.append(classes.index(name))
    return torch.tensor(boxes, dtype=torch.float32).reshape(-1, 9), torch.tensor(labels, dtype=torch.int64)
