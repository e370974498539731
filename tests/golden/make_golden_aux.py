#!/usr/bin/env python3
"""Golden fixtures of the rows either side of the hot path, produced by EXECUTING THE
REFERENCE'S OWN PYTHON in the build container (needs /root/reference, which never travels to the GPU box).

* losses.npz   ``losses.focal.FocalLoss`` (losses/focal.py:12-90, losses/_functional.py:37-108; pure torch, imported
               unmodified): loss values and input gradients for multiclass / binary / multilabel modes, with and
               without ``ignore_index``, 'mean' and 'sum'.

* kitti_eval.npz  the KITTI-AP evaluator: ``evaluators/kitti_utils/eval.py`` (kitti_eval, eval_class,
               d3_box_overlap), ``rotate_iou.py`` (rotate_iou_gpu_eval) and the label reader of ``kitti_common.py:561-669``
               executed on a synthetic label set.  numba is absent: an identity ``jit`` runs the jitted functions as the
               plain Python they are, and the numba.cuda kernel is run thread by thread (see ``_numba_stub``).

* result2kitti.npz  ``result2kitti`` (evaluators/result2kitti.py:212-268) on synthetic calibration files and a results JSON.

Outputs are DATA ONLY (inputs and expected outputs); no reference source text is stored.

    python tests/golden/make_golden_aux.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("SGV3D_GOLDEN_OUT", HERE)      # tests/test_golden_regen_cpu.py regenerates into a temp dir
REF = "/root/reference"


def make_losses():
    sys.path.insert(0, REF)
    from losses.focal import FocalLoss                        # the reference's class
    g = torch.Generator().manual_seed(7)
    out = {}
    cases = [
        ("mc_mean", dict(mode='multiclass', alpha=0.25, gamma=2, reduction="mean"), (3, 7, 6, 10)),     # the BSM config (:249)
        ("mc_sum_g15", dict(mode='multiclass', alpha=0.4, gamma=1.5, reduction="sum"), (2, 5, 4, 6)),
        ("mc_ignore", dict(mode='multiclass', alpha=None, gamma=2.0, ignore_index=3, reduction="mean"), (2, 4, 5, 7)),
        ("bin_mean", dict(mode='binary', alpha=0.25, gamma=2.0, reduction="mean"), (4, 1, 5, 6)),
        ("ml_sum_g1", dict(mode='multilabel', alpha=0.6, gamma=1.0, reduction="sum"), (2, 3, 4, 5)),
    ]
    for name, kw, shape in cases:
        x = (torch.randn(shape, generator=g, dtype=torch.float64) * 3).requires_grad_(True)
        if kw['mode'] == 'multiclass':
            y = torch.randint(0, shape[1], (shape[0],) + shape[2:], generator=g)
        else:
            y = torch.randint(0, 2, shape, generator=g)
        loss = FocalLoss(**kw)(x, y)
        loss.backward()
        out[name + "_x"] = x.detach().numpy()
        out[name + "_y"] = y.numpy()
        out[name + "_loss"] = np.float64(loss.item())
        out[name + "_grad"] = x.grad.numpy()
        out[name + "_kw"] = np.array(repr(kw))
        # the same inputs in float32 (what the harness feeds), value only
        out[name + "_loss32"] = np.float32(FocalLoss(**kw)(x.detach().float(), y).item())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("losses.npz:", sorted(k for k in out if k.endswith("_loss")))




# ------------------------------------------------------------------------------------------------ KITTI evaluator
def _numba_stub():
    """``numba`` is absent here; the reference's jitted functions are plain Python underneath, so an identity ``jit``
    runs them unmodified.  The ``numba.cuda`` kernel of rotate_iou.py is executed block by block, thread by thread:
    every block runs twice (the second pass sees the shared arrays fully loaded -- the kernel only stores results, so
    repeating it is harmless), which stands in for ``cuda.syncthreads()``."""
    import types
    nb = types.ModuleType('numba')

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = jit
    nb.prange = range
    nb.float32 = np.float32
    nb.int32 = np.int32
    cuda = types.ModuleType('numba.cuda')
    state = types.SimpleNamespace(shared=None, k=0)
    cuda.blockIdx = types.SimpleNamespace(x=0, y=0)
    cuda.threadIdx = types.SimpleNamespace(x=0)

    class _Local:
        @staticmethod
        def array(shape, dtype):
            return np.zeros(shape, dtype=np.float32)

    class _Shared:
        @staticmethod
        def array(shape, dtype):
            i = state.k
            state.k += 1
            if i >= len(state.shared):
                state.shared.append(np.zeros(shape, dtype=np.float32))
            return state.shared[i]
    cuda.local = _Local
    cuda.shared = _Shared
    cuda.syncthreads = lambda: None
    cuda.select_device = lambda i: None

    class _Stream:
        def auto_synchronize(self):
            import contextlib
            return contextlib.nullcontext()
    cuda.stream = lambda: _Stream()

    class _Dev:
        def __init__(self, a):
            self.a = a

        def copy_to_host(self, out, stream=None):
            out[...] = self.a
    cuda.to_device = lambda a, stream=None: _Dev(a)

    def cjit(*a, **k):
        device = k.get('device', False)

        def deco(f):
            if device:
                return f

            class _Kernel:
                def __getitem__(self, cfg):
                    grid, block = cfg[0], cfg[1]

                    def launch(*args):
                        args = [x.a if isinstance(x, _Dev) else x for x in args]
                        for bx in range(int(grid[0])):
                            for by in range(int(grid[1])):
                                state.shared = []
                                for _ in range(2):
                                    for tx in range(int(block)):
                                        cuda.blockIdx.x, cuda.blockIdx.y, cuda.threadIdx.x = bx, by, tx
                                        state.k = 0
                                        f(*args)
                    return launch
            return _Kernel()
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return deco
    cuda.jit = cjit
    nb.cuda = cuda
    sys.modules['numba'] = nb
    sys.modules['numba.cuda'] = cuda


def _label_line(name, trunc, occ, alpha, bbox, hwl, loc, ry, score=None):
    f = [name, f'{trunc:.2f}', str(int(occ)), f'{alpha:.4f}'] + [f'{v:.2f}' for v in bbox] + [f'{v:.4f}' for v in hwl] + \
        [f'{v:.4f}' for v in loc] + [f'{ry:.4f}']
    if score is not None:
        f.append(f'{score:.4f}')
    return ' '.join(f)


def synth_kitti_set(rng, n_img=40):
    """Label-file TEXT (KITTI format) for ground truth and detections of a small synthetic set that exercises every
    branch of the evaluator: four label names (Bus is mapped to Car by the reader), all occlusion / truncation levels,
    2-D heights around the difficulty limits, detections = perturbed copies of the ground truth + false positives +
    duplicates, images without ground truth or without detections."""
    gts, dts = [], []
    for i in range(n_img):
        g_lines, d_lines = [], []
        n = (int(rng.integers(0, 9)) if i else 5) if i != 3 else 0     # the first file must hold an object (eval.py:705-708 indexes it)
        for k in range(n):
            name = ['Car', 'Pedestrian', 'Cyclist', 'Bus'][int(rng.integers(0, 4))]
            l, w, h = {'Car': (4.3, 1.9, 1.6), 'Bus': (10.5, 2.6, 3.1), 'Pedestrian': (0.6, 0.6, 1.7), 'Cyclist': (1.7, 0.6, 1.6)}[name]
            l, w, h = l * rng.uniform(0.85, 1.15), w * rng.uniform(0.85, 1.15), h * rng.uniform(0.9, 1.1)
            x, y, z = rng.uniform(-30, 30), rng.uniform(1.0, 2.5), rng.uniform(10, 90)
            ry = rng.uniform(-np.pi, np.pi)
            hh = rng.choice([20.0, 30.0, 45.0, 80.0], p=[0.1, 0.15, 0.35, 0.4])
            u, v = rng.uniform(100, 1700), rng.uniform(200, 900)
            bbox = (u, v, u + hh * 1.6, v + hh)
            occ, trunc = int(rng.choice(4, p=[0.55, 0.2, 0.15, 0.1])), float(rng.choice([0.0, 0.1, 0.25, 0.45, 0.7], p=[0.5, 0.2, 0.1, 0.1, 0.1]))
            g_lines.append(_label_line(name, trunc, occ, rng.uniform(-3, 3), bbox, (h, w, l), (x, y, z), ry))
            if rng.uniform() < 0.8:                                            # a detection near this object
                s = rng.uniform(0.05, 1.0)
                jit = lambda sc: rng.normal(0, sc)
                dn = name if name != 'Bus' else 'Car'
                if rng.uniform() < 0.1:
                    dn = ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(0, 3))]
                db = tuple(b + jit(2.0) for b in bbox)
                d_lines.append(_label_line(dn, 0.0, 0, rng.uniform(-3, 3), db, (h * (1 + jit(.05)), w * (1 + jit(.05)), l * (1 + jit(.05))),
                                           (x + jit(.04 * l), y + jit(.03 * h), z + jit(.04 * l)), ry + jit(0.08), s))
                if rng.uniform() < 0.15:                                       # duplicate with a lower score
                    d_lines.append(_label_line(dn, 0.0, 0, 0.1, db, (h, w, l), (x + jit(.06 * l), y, z + jit(.06 * l)), ry, s * 0.6))
        for k in range(int(rng.integers(0, 4))):                               # false positives
            dn = ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(0, 3))]
            u, v = rng.uniform(100, 1700), rng.uniform(200, 900)
            hh = rng.choice([18.0, 35.0, 60.0])
            d_lines.append(_label_line(dn, 0.0, 0, 0.0, (u, v, u + 50, v + hh), (1.6, 1.8, 4.2),
                                       (rng.uniform(-30, 30), 1.5, rng.uniform(10, 90)), rng.uniform(-3, 3), rng.uniform(0.05, 0.9)))
        if i == 5:
            d_lines = []
        gts.append('\n'.join(g_lines) + ('\n' if g_lines else ''))
        dts.append('\n'.join(d_lines) + ('\n' if d_lines else ''))
    return gts, dts


def make_kitti_eval():
    import tempfile
    import types
    _numba_stub()
    sk = types.ModuleType('skimage')
    sk.io = types.ModuleType('skimage.io')
    sys.modules['skimage'] = sk
    sys.modules['skimage.io'] = sk.io
    sys.path.insert(0, REF)
    import importlib.util

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    pkg = types.ModuleType('evaluators')
    pkg.__path__ = [os.path.join(REF, 'evaluators')]
    sys.modules['evaluators'] = pkg
    sub = types.ModuleType('evaluators.kitti_utils')
    sub.__path__ = [os.path.join(REF, 'evaluators', 'kitti_utils')]
    sys.modules['evaluators.kitti_utils'] = sub
    riou = load('evaluators.kitti_utils.rotate_iou', 'evaluators/kitti_utils/rotate_iou.py')
    ev = load('evaluators.kitti_utils.eval', 'evaluators/kitti_utils/eval.py')
    kc = load('evaluators.kitti_utils.kitti_common', 'evaluators/kitti_utils/kitti_common.py')
    rng = np.random.default_rng(11)
    out = {}
    # ---- rotated IoU matrices (rotate_iou.py:340-378) and the 3-D overlap built on them (eval.py:120-160)
    def rboxes(n):
        return np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(0.5, 8, n), rng.uniform(0.5, 4, n),
                         rng.uniform(-np.pi, np.pi, n)], 1)
    b, q = rboxes(70), rboxes(37)
    q[:5] = b[:5]                                                   # identical boxes
    q[5:9, :2] = b[5:9, :2]; q[5:9, 4] = b[5:9, 4] + np.pi / 2      # same centre, turned by 90 degrees
    q[9:12] = b[9:12]; q[9:12, 2:4] *= 0.5                          # contained boxes
    b[20:24, 4] = 0.0; q[12:16, 4] = 0.0                            # axis-aligned pairs
    out['riou_boxes'], out['riou_qboxes'] = b, q
    for c in (-1, 0, 1, 2):
        out[f'riou_c{c}'] = riou.rotate_iou_gpu_eval(b, q, c)
    def boxes7(n):
        return np.concatenate([rng.uniform(-10, 10, (n, 1)), rng.uniform(0.5, 3, (n, 1)), rng.uniform(5, 25, (n, 1)),
                               rng.uniform(0.5, 6, (n, 3)), rng.uniform(-np.pi, np.pi, (n, 1))], 1)
    b7, q7 = boxes7(40), boxes7(33)
    q7[:6] = b7[:6] + rng.normal(0, 0.05, (6, 7))
    out['d3_boxes'], out['d3_qboxes'] = b7, q7
    out['d3_overlap'] = ev.d3_box_overlap(b7, q7).astype(np.float64)
    # ---- the whole evaluation through the label-file reader
    gts, dts = synth_kitti_set(rng)
    out['label_gt'] = np.array(gts)
    out['label_dt'] = np.array(dts)
    with tempfile.TemporaryDirectory() as d:
        for sub_, texts in (('gt', gts), ('dt', dts)):
            os.makedirs(os.path.join(d, sub_))
            for i, t in enumerate(texts):
                with open(os.path.join(d, sub_, f'{i:06d}.txt'), 'w') as f:
                    f.write(t)
        dt_annos, ids = kc.get_label_annos(os.path.join(d, 'dt'), return_ids=True)
        gt_annos = kc.get_label_annos(os.path.join(d, 'gt'), image_ids=ids)
    result, ret = ev.kitti_eval(gt_annos, dt_annos, ["Car", "Pedestrian", "Cyclist"], metric="R40")
    out['result_text'] = np.array(result)
    out['ret_keys'] = np.array(sorted(ret))
    out['ret_vals'] = np.array([ret[k] for k in sorted(ret)], np.float64)
    result11, ret11 = ev.kitti_eval(gt_annos, dt_annos, ["Car", "Pedestrian", "Cyclist"], eval_types=['bbox', 'bev', '3d'], metric="R11")
    out['ret11_vals'] = np.array([ret11[k] for k in sorted(ret11)], np.float64)
    # precision / recall / orientation curves of one metric each (eval_class, eval.py:441-572)
    mo = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.7, 0.5, 0.5], [0.5, 0.25, 0.25], [0.5, 0.25, 0.25]])], 0)
    for metric in (0, 1, 2):
        r = ev.eval_class(gt_annos, dt_annos, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        out[f'curve{metric}_precision'] = r['precision']
        out[f'curve{metric}_recall'] = r['recall']
        if metric == 0:
            out['curve0_orientation'] = r['orientation']
    a = kc.get_label_anno  # reader output of the first non-empty ground-truth file (names after the Bus -> Car mapping)
    first = next(i for i, t in enumerate(gts) if t)
    with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
        f.write(gts[first])
    an = a(f.name)
    os.unlink(f.name)
    out['reader_index'] = np.int64(first)
    for k in ('truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score'):
        out['reader_' + k] = np.asarray(an[k], np.float64)
    out['reader_name'] = np.array([str(s) for s in an['name']])
    np.savez_compressed(os.path.join(OUT, "kitti_eval.npz"), **out)
    print("kitti_eval.npz:", result[:400])


# ------------------------------------------------------------------------------------------------ result2kitti
def make_result2kitti():
    """evaluators/result2kitti.py:212-268 (``result2kitti``, the KITTI-format DAIR-V2X-I / Rope3D roots) executed on
    synthetic calibration files and a synthetic results JSON -> the label files it writes."""
    import json
    import tempfile
    import types
    _numba_stub()
    for name in ('cv2', 'mmcv', 'skimage', 'skimage.io'):
        sys.modules.setdefault(name, types.ModuleType(name))
    pq = types.ModuleType('pyquaternion')
    pq.Quaternion = object
    sys.modules.setdefault('pyquaternion', pq)
    sys.path.insert(0, REF)
    import importlib
    r2k = importlib.import_module('evaluators.result2kitti')
    rng = np.random.default_rng(5)
    calibs, results = {}, {}
    for sid in (3, 17, 250):
        pitch = np.deg2rad(rng.uniform(8, 14))
        # lidar/ego (x forward, y left, z up) -> camera (x right, y down, z forward), pitched down
        base = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], np.float64)
        rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        R_ = rx @ base
        t_ = np.array([rng.uniform(-0.2, 0.2), rng.uniform(5.0, 7.0), rng.uniform(-0.3, 0.3)])
        tr = np.concatenate([R_, t_[:, None]], 1)
        P2 = np.array([[2183.4, 0, 940.6, 0], [0, 2329.3, 567.6, 0], [0, 0, 1, 0]])
        calibs[sid] = "P2: " + " ".join(f"{v:.6f}" for v in P2.reshape(-1)) + "\n" + \
                      "Tr_velo_to_cam: " + " ".join(f"{v:.8f}" for v in tr.reshape(-1)) + "\n"
        preds = []
        for k in range(int(rng.integers(3, 9))):
            name = ['car', 'van', 'truck', 'bus', 'pedestrian', 'bicycle', 'trailer', 'motorcycle', 'barrier'][int(rng.integers(0, 9))]
            preds.append(dict(translation=[float(rng.uniform(15, 90)), float(rng.uniform(-20, 20)), float(rng.uniform(-1.5, -0.5))],
                              size=[float(rng.uniform(0.5, 2.5)), float(rng.uniform(0.5, 10)), float(rng.uniform(1.4, 3.2))],
                              box_yaw=float(rng.uniform(-np.pi, np.pi)), detection_score=float(rng.uniform(0.2, 1.0)),
                              detection_name=name))
        results[f"training/image_2/{sid:06d}.jpg"] = preds
    out = {'calib_ids': np.array(sorted(calibs)), 'calib_text': np.array([calibs[k] for k in sorted(calibs)]),
           'results_json': np.array(json.dumps({'meta': {}, 'results': results}))}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'root', 'training', 'calib'))
        for sid, text in calibs.items():
            with open(os.path.join(d, 'root', 'training', 'calib', f'{sid:06d}.txt'), 'w') as f:
                f.write(text)
        rf = os.path.join(d, 'results_nusc.json')
        with open(rf, 'w') as f:
            json.dump({'meta': {}, 'results': results}, f)
        path = r2k.result2kitti(rf, os.path.join(d, 'out'), os.path.join(d, 'root'), os.path.join(d, 'gt'), demo=False)
        out['label_text'] = np.array([open(os.path.join(path, f'{sid:06d}.txt')).read() for sid in sorted(calibs)])
    # the same detections through result2kitti_dair (:270-328): raw DAIR-V2X-I root, JSON calibration files
    dair_cam, dair_v2c = {}, {}
    for sid in sorted(calibs):
        rows = calibs[sid].splitlines()
        P2 = np.array([float(v) for v in rows[0].split(' ')[1:]]).reshape(3, 4)
        tr = np.array([float(v) for v in rows[1].split(' ')[1:]]).reshape(3, 4)
        dair_cam[sid] = json.dumps({"cam_D": [0.0] * 5, "cam_K": P2[:, :3].reshape(-1).tolist()})
        dair_v2c[sid] = json.dumps({"rotation": tr[:, :3].tolist(), "translation": tr[:, 3:].tolist()})
    out['dair_cam_json'] = np.array([dair_cam[k] for k in sorted(calibs)])
    out['dair_v2c_json'] = np.array([dair_v2c[k] for k in sorted(calibs)])
    with tempfile.TemporaryDirectory() as d:
        for sub_ in ('camera_intrinsic', 'virtuallidar_to_camera'):
            os.makedirs(os.path.join(d, 'root', 'calib', sub_))
        for sid in calibs:
            open(os.path.join(d, 'root', 'calib', 'camera_intrinsic', f'{sid:06d}.json'), 'w').write(dair_cam[sid])
            open(os.path.join(d, 'root', 'calib', 'virtuallidar_to_camera', f'{sid:06d}.json'), 'w').write(dair_v2c[sid])
        rf = os.path.join(d, 'results_nusc.json')
        with open(rf, 'w') as f:
            json.dump({'meta': {}, 'results': results}, f)
        path = r2k.result2kitti_dair(rf, os.path.join(d, 'out'), os.path.join(d, 'root'), os.path.join(d, 'gt'), demo=False)
        out['dair_label_text'] = np.array([open(os.path.join(path, f'{sid:06d}.txt')).read() for sid in sorted(calibs)])
    # the same detections through result2kitti_rope3d (:330-393): raw Rope3D root -- per-image ground-plane ("denorm")
    # files, KITTI-style calib files named by token, and a token -> sample-id map read from a path relative to the
    # working directory.  cv2 is absent: cv2.Rodrigues (scripts/gen_info_rope3d.py:73) is supplied by scipy's
    # rotation-vector -> matrix conversion (the same Rodrigues formula, float64).
    from scipy.spatial.transform import Rotation
    sys.modules['cv2'].Rodrigues = lambda v: (Rotation.from_rotvec(np.asarray(v, np.float64)).as_matrix(), None)
    r2k.cv2 = sys.modules['cv2']
    tokens = {3: "1632_fa2sd4a11North151_420_1613710840_1613716786_1_obstacle", 17: "val_cam07_000017", 250: "tok250"}
    split = {3: "training", 17: "validation", 250: "training"}
    denorms, rope_calib, rope_results = {}, {}, {}
    for sid in sorted(calibs):
        pitch = np.deg2rad(rng.uniform(8, 14))
        roll = np.deg2rad(rng.uniform(-1.5, 1.5))
        n = np.array([np.sin(roll), -np.cos(pitch) * np.cos(roll), -np.sin(pitch) * np.cos(roll)])   # ground normal in the camera frame
        denorms[sid] = " ".join(f"{v:.10f}" for v in list(n) + [float(rng.uniform(5.0, 7.5))]) + "\n"
        rope_calib[sid] = calibs[sid].splitlines()[0] + "\n"
        rope_results[tokens[sid]] = results[f"training/image_2/{sid:06d}.jpg"]
    out['rope_tokens'] = np.array([tokens[k] for k in sorted(calibs)])
    out['rope_split'] = np.array([split[k] for k in sorted(calibs)])
    out['rope_denorm_text'] = np.array([denorms[k] for k in sorted(calibs)])
    out['rope_calib_text'] = np.array([rope_calib[k] for k in sorted(calibs)])
    out['rope_results_json'] = np.array(json.dumps({'meta': {}, 'results': rope_results}))
    with tempfile.TemporaryDirectory() as d:
        for sid in calibs:
            for sub_, text in (('denorm', denorms[sid]), ('calib', rope_calib[sid])):
                os.makedirs(os.path.join(d, 'root', split[sid], sub_), exist_ok=True)
                open(os.path.join(d, 'root', split[sid], sub_, tokens[sid] + '.txt'), 'w').write(text)
        os.makedirs(os.path.join(d, 'data', 'rope3d-kitti'))
        json.dump({tokens[k]: f"{k:06d}" for k in calibs}, open(os.path.join(d, 'data', 'rope3d-kitti', 'map_token2id.json'), 'w'))
        rf = os.path.join(d, 'results_nusc.json')
        json.dump({'meta': {}, 'results': rope_results}, open(rf, 'w'))
        cwd = os.getcwd()
        os.chdir(d)
        try:
            path = r2k.result2kitti_rope3d(rf, os.path.join(d, 'out'), os.path.join(d, 'root'), os.path.join(d, 'gt'), demo=False)
        finally:
            os.chdir(cwd)
        out['rope_label_text'] = np.array([open(os.path.join(path, f'{sid:06d}.txt')).read() for sid in sorted(calibs)])
    np.savez_compressed(os.path.join(OUT, "result2kitti.npz"), **out)
    print("result2kitti.npz:", out['label_text'][0][:300])
    print("rope3d:", out['rope_label_text'][0][:300])


if __name__ == "__main__":
    which = sys.argv[1:] or ["losses", "kitti_eval", "result2kitti"]
    if "losses" in which:
        make_losses()
    if "kitti_eval" in which:
        make_kitti_eval()
    if "result2kitti" in which:
        make_result2kitti()
