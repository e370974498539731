#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
Third-party packages the reference imports but that are absent here (mmcv, mmdet, mmdet3d, cv2) are
replaced by empty ``sys.modules`` stubs; none of the captured code paths touches them.  Executed
unmodified from the reference:

* ``LSSFPN.create_frustum``      layers/backbones/lss_fpn.py:325-348
* ``LSSFPN.get_geometry`` / ``height2localtion``   lss_fpn.py:350-401
* the quantise expression       lss_fpn.py:487-488   (int cast emulated with GPU semantics, see below)
* ``VoxelPooling.apply`` fwd+bwd ops/voxel_pooling/voxel_pooling.py:10-69 around a stub
  ``voxel_pooling_ext`` that follows voxel_pooling_forward_cuda.cu:9-36 literally (the CUDA
  extension itself cannot be built or loaded here).

* the dataset-side geometry helpers (get_denorm, get_sensor2virtual, get_reference_height, the ida /
  bda matrices)   dataset/nusc_mv_det_dataset.py:41-179,433-446   -> input_contract.npz

Outputs are DATA ONLY (inputs and expected outputs); no reference source text is stored.

    python tests/golden/make_golden.py
"""
import hashlib
import math
import os
import sys
import types
import warnings

import numpy as np
import torch
from torch import nn

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("SGV3D_GOLDEN_OUT", HERE)      # tests/test_golden_regen_cpu.py regenerates into a temp dir
REF = "/root/reference"
f32 = np.float32


# ------------------------------------------------------------------ stub import of the reference
def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Dummy(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def _raise(*a, **k):
    raise RuntimeError("third-party stub called")


def _kernel_stub(batch_size, num_points, num_channels, nvx, nvy, nvz, geom, feats, out, pos_memo):
    """voxel_pooling_forward_cuda.cu:9-36, one point at a time, ascending order."""
    nvx, nvy, nvz = int(nvx), int(nvy), int(nvz)
    g = geom.reshape(-1, 3)
    f = feats.reshape(-1, num_channels)
    o = out.view(-1, num_channels)
    pm = pos_memo.view(-1, 3)
    for pt in range(batch_size * num_points):
        b = pt // num_points
        x, y, z = int(g[pt, 0]), int(g[pt, 1]), int(g[pt, 2])
        if x < 0 or x >= nvx or y < 0 or y >= nvy or z < 0 or z >= nvz:
            continue
        pm[pt, 0] = b
        pm[pt, 1] = y
        pm[pt, 2] = x
        o[(b * nvy + y) * nvx + x] += f[pt]
    return 1


def import_reference():
    _stub('mmcv')
    _stub('mmcv.cnn', build_conv_layer=_raise)
    _stub('mmdet')
    _stub('mmdet.models', build_backbone=_raise)
    _stub('mmdet.models.backbones')
    _stub('mmdet.models.backbones.resnet', BasicBlock=_Dummy)
    _stub('mmdet.core', reduce_mean=_raise)
    _stub('mmdet3d')
    _stub('mmdet3d.models', build_neck=_raise)
    _stub('mmdet3d.core', draw_heatmap_gaussian=_raise, gaussian_radius=_raise)
    _stub('mmdet3d.models.dense_heads')
    _stub('mmdet3d.models.dense_heads.centerpoint_head', CenterHead=_Dummy)
    _stub('mmdet3d.models.utils', clip_sigmoid=_raise)
    _stub('cv2')
    sys.path.insert(0, REF)
    ext = types.ModuleType('ops.voxel_pooling.voxel_pooling_ext')
    ext.voxel_pooling_forward_wrapper = _kernel_stub
    sys.modules['ops.voxel_pooling.voxel_pooling_ext'] = ext
    import layers.backbones.lss_fpn as L
    import ops.voxel_pooling  # noqa: F401  (its __init__ rebinds the name to the function)
    VP = sys.modules['ops.voxel_pooling.voxel_pooling']
    return L, VP


def make_lss(L, final_dim, downsample, d_bound, xb, yb, zb):
    """An LSSFPN instance with only the geometry buffers (no third-party sub-modules)."""
    obj = L.LSSFPN.__new__(L.LSSFPN)
    nn.Module.__init__(obj)
    obj.downsample_factor = downsample
    obj.d_bound = d_bound
    obj.final_dim = final_dim
    rows = [xb, yb, zb]
    obj.register_buffer('voxel_size', torch.Tensor([r[2] for r in rows]))                    # :281-283
    obj.register_buffer('voxel_coord', torch.Tensor([r[0] + r[2] / 2.0 for r in rows]))      # :284-288
    obj.register_buffer('voxel_num', torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows]))  # :289-292
    obj.register_buffer('frustum', obj.create_frustum())                                     # :293
    return obj


# ------------------------------------------------------------------ synthetic DAIR-like calibration
def _rodrigues(rvec):
    th = float(np.linalg.norm(rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


def _equation_plane(p):
    (x1, y1, z1), (x2, y2, z2), (x3, y3, z3) = p
    a1, b1, c1 = x2 - x1, y2 - y1, z2 - z1
    a2, b2, c2 = x3 - x1, y3 - y1, z3 - z1
    a = b1 * c2 - b2 * c1
    b = a2 * c1 - a1 * c2
    c = a1 * b2 - b1 * a2
    return np.array([a, b, c, -a * x1 - b * y1 - c * z1])


def make_calib(pitch_deg=11.0, cam_h=5.5, yaw_deg=0.0, roll_deg=0.0,
               fx=2183.375, fy=2329.2976, cx=940.59, cy=567.568, resize=0.8, crop=(0.0, 0.0)):
    """Camera (x right, y down, z forward) looking along ego +x, pitched down; matrices built with
    the dataset's formulas (dataset/nusc_mv_det_dataset.py:63-86, 433-446) restated in float64."""
    p, yw, rl = (math.radians(a) for a in (pitch_deg, yaw_deg, roll_deg))
    fwd = np.array([math.cos(p), 0.0, -math.sin(p)])
    right = np.array([0.0, -1.0, 0.0])
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    Rroll = _rodrigues(np.array([0.0, 0.0, 1.0]) * rl)           # about the optical axis
    Rz = np.array([[math.cos(yw), -math.sin(yw), 0], [math.sin(yw), math.cos(yw), 0], [0, 0, 1]])
    s2e = np.eye(4)
    s2e[:3, :3] = Rz @ R @ Rroll
    s2e[:3, 3] = [0.0, 0.0, cam_h]
    e2s = np.linalg.inv(s2e)
    gp = np.array([[0, 0, 0, 1.0], [0, 1, 0, 1], [1, 1, 0, 1]])
    denorm = -1 * _equation_plane((e2s @ gp.T).T[:, :3])
    origin = np.array([0.0, 1.0, 0.0])
    target = -1 * denorm[:3]
    tn = target / np.linalg.norm(target)
    sita = math.acos(float(np.clip(np.inner(tn, origin), -1, 1)))
    nv = np.cross(tn, origin)
    s2v = np.eye(4)
    if np.linalg.norm(nv) > 1e-12:
        nv = (nv / np.linalg.norm(nv)).astype(f32).astype(np.float64)
        s2v[:3, :3] = _rodrigues(nv * sita).astype(f32)
    refh = f32(abs(denorm[3]) / np.linalg.norm(denorm[:3]))
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    ida = np.eye(4)
    ida[0, 0] = ida[1, 1] = resize
    ida[0, 3], ida[1, 3] = -crop[0], -crop[1]
    return dict(sensor2ego=s2e.astype(f32), sensor2virtual=s2v.astype(f32), intrin=K.astype(f32),
                ida=ida.astype(f32), bda=np.eye(4, dtype=f32), reference_height=refh)


CALIBS = {
    # name: kwargs
    "dair_p11_h5.5": dict(),
    "p5_h8_yaw3": dict(pitch_deg=5.0, cam_h=8.0, yaw_deg=3.0),
    "p20_h4_roll2": dict(pitch_deg=20.0, cam_h=4.0, roll_deg=2.0, crop=(16.0, 8.0)),
    "p14_h6.3_yaw-7_roll-1": dict(pitch_deg=14.0, cam_h=6.3, yaw_deg=-7.0, roll_deg=-1.0,
                                  fx=2100.0, fy=2250.5, cx=955.25, cy=540.75),
    # full-resolution frame (no resize): what a 1080x1920 image padded to 1088x1920 (BASELINE cfg-3) carries
    "p11_h5.5_fullres": dict(resize=1.0),
}


def cvt_i32_gpu(x):
    """float32 -> int32 with the GPU semantics the reference runs with (.int() on a CUDA tensor:
    truncate toward zero, saturate, NaN -> 0); torch-CPU's cvttss2si differs only on NaN/inf/overflow
    (SURVEY.md §7(a)), and the two are asserted equal on every finite in-range value below."""
    x = np.asarray(x, f32)
    out = np.zeros(x.shape, np.int32)
    nan = np.isnan(x)
    hi = x >= f32(2147483648.0)
    lo = x <= f32(-2147483648.0)
    ok = ~(nan | hi | lo)
    out[ok] = np.trunc(x[ok]).astype(np.int32)
    out[hi] = 2147483647
    out[lo] = -2147483648
    return out


def run_geometry(obj, c):
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).view(1, 1, 4, 4)
    with torch.no_grad():
        geom = obj.get_geometry(T(c['sensor2ego']), T(c['sensor2virtual']), T(c['intrin']), T(c['ida']),
                                torch.tensor([[float(c['reference_height'])]]),
                                torch.from_numpy(c['bda']).view(1, 4, 4))
        q = ((geom - (obj.voxel_coord - obj.voxel_size / 2.0)) / obj.voxel_size)      # :487-488 (float part)
        cpu_int = q.int().numpy()[0, 0]
        # the reference's own 4x4 preparation, same torch calls and shapes as lss_fpn.py:361,367,390
        ida_inv = T(c['ida']).view(1, 1, 1, 1, 1, 4, 4).inverse()[0, 0, 0, 0, 0].numpy()
        cv = T(c['sensor2virtual']).matmul(torch.inverse(T(c['intrin'])))[0, 0].numpy()
        ce = T(c['sensor2ego']).matmul(torch.inverse(T(c['sensor2virtual'])))[0, 0].numpy()
    geom = geom.numpy()[0, 0]
    gi = cvt_i32_gpu(q.numpy()[0, 0])
    finite = np.isfinite(q.numpy()[0, 0]) & (np.abs(q.numpy()[0, 0]) < 2.0e9)
    assert np.array_equal(gi[finite], cpu_int[finite])
    return geom, gi, ida_inv, cv, ce


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    L, VP = import_reference()
    torch.manual_seed(0)
    out = {}

    # ---------------- G1: frustum for every (final_dim, downsample, d_bound) the exps use ------
    fr_cfgs = {
        "r50_864x1536_s16_d90": ((864, 1536), 16, [-2.0, 0.0, 90]),
        "bsm_864x1536_s8_d180": ((864, 1536), 8, [-2.0, 3.5, 180]),
        "rope_864x1536_s16_d90": ((864, 1536), 16, [-0.5, 2.5, 90]),
        "cfg3_1088x1920_s16_d90": ((1088, 1920), 16, [-2.0, 0.0, 90]),
        "small_80x112_s16_d6": ((80, 112), 16, [-2.0, 0.0, 6]),
    }
    fr_out = {}
    for name, (fd, ds, db) in fr_cfgs.items():
        obj = make_lss(L, fd, ds, db, [0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
        fr = obj.frustum.numpy()
        # the tensor is separable: store its three axes + a hash of the whole thing
        fr_out[name + "/xs"] = fr[0, 0, :, 0].copy()
        fr_out[name + "/ys"] = fr[0, :, 0, 1].copy()
        fr_out[name + "/ds"] = fr[:, 0, 0, 2].copy()
        fr_out[name + "/shape"] = np.array(fr.shape)
        fr_out[name + "/sha256"] = np.frombuffer(bytes.fromhex(sha(fr)), np.uint8)
        fr_out[name + "/cfg"] = np.array([fd[0], fd[1], ds, db[0], db[1], db[2]], np.float64)
        assert np.all(fr[..., 3] == 1)
    np.savez_compressed(os.path.join(OUT, "frustum.npz"), **fr_out)

    # ---------------- G2/G3: geometry, small exact tensors + full-size hashes -------------------
    geo = {}
    small = make_lss(L, (80, 112), 16, [-2.0, 0.0, 6], [0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
    full = make_lss(L, (864, 1536), 16, [-2.0, 0.0, 90], [0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
    full128 = make_lss(L, (864, 1536), 16, [-2.0, 0.0, 90], [0, 102.4, 0.8], [-51.2, 51.2, 0.8], [-5, 3, 8])
    # BASELINE cfg-3 (R101, 1088x1920 padded frame, 0.2 m cells -> 512x512 BEV, N = 734 400 points) and
    # cfg-5 (SGV3D BSM: stride-8 frustum 108x192, 180 height bins in [-2, 3.5] -> N = 3 732 480 points)
    cfg3 = make_lss(L, (1088, 1920), 16, [-2.0, 0.0, 90], [0, 102.4, 0.2], [-51.2, 51.2, 0.2], [-5, 3, 8])
    cfg5 = make_lss(L, (864, 1536), 8, [-2.0, 3.5, 180], [0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
    calibs = {k: make_calib(**kw) for k, kw in CALIBS.items()}
    # a ray exactly parallel to the ground: pitch 0, principal point on a frustum row of the small
    # config (ys[2] = 39.5 -> v' = 39.5/0.8 = 49.375): ratio = h/0 = inf, 0*inf = NaN
    calibs["nan_ray_small"] = make_calib(pitch_deg=0.0, cam_h=5.0, fx=100.0, fy=100.0, cx=70.0, cy=49.375)
    for name, c in calibs.items():
        for key in ("sensor2ego", "sensor2virtual", "intrin", "ida", "bda"):
            geo[f"{name}/{key}"] = c[key]
        geo[f"{name}/reference_height"] = np.array(c["reference_height"], f32)
        g, gi, ida_inv, cv, ce = run_geometry(small, c)
        geo[f"{name}/small/geom"] = g
        geo[f"{name}/small/geom_xyz"] = gi
        geo[f"{name}/ref_ida_inv"] = ida_inv
        geo[f"{name}/ref_combine_virtual"] = cv
        geo[f"{name}/ref_combine_ego"] = ce
        if name == "nan_ray_small":
            assert np.isnan(g).any(), "fixture must contain a NaN ray"
            continue
        for tag, obj in (("full256", full), ("full128", full128), ("cfg3_512", cfg3), ("cfg5_s8d180", cfg5)):
            g, gi, *_ = run_geometry(obj, c)
            vn = obj.voxel_num.numpy()
            inr = ((gi[..., 0] >= 0) & (gi[..., 0] < vn[0]) & (gi[..., 1] >= 0) & (gi[..., 1] < vn[1])
                   & (gi[..., 2] >= 0) & (gi[..., 2] < vn[2]))
            lin = (gi[..., 1].astype(np.int64) * vn[0] + gi[..., 0])[inr]
            cnt = np.bincount(lin, minlength=int(vn[0] * vn[1]))
            geo[f"{name}/{tag}/geom_xyz_sha256"] = np.frombuffer(bytes.fromhex(sha(gi)), np.uint8)
            geo[f"{name}/{tag}/stats"] = np.array(
                [inr.mean(), (cnt > 0).sum(), cnt[cnt > 0].mean(), cnt.max()], np.float64)
            # strided sample of the index tensor (every 7th height bin, 5th row, 9th column)
            geo[f"{name}/{tag}/geom_xyz_sample"] = gi[::7, ::5, ::9].copy()
            print(f"{name:24s} {tag}: in-range {inr.mean():.4f} hit voxels {(cnt > 0).sum()} "
                  f"mean {cnt[cnt > 0].mean():.2f} max {cnt.max()}")
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **geo)

    # ---------------- VP-H/VP-K/VP-B: the reference's autograd wrapper around the literal stub ---
    vp = {}
    gen = torch.Generator().manual_seed(1234)
    cases = {
        # name: (B, dims (D,H,W), C, voxel_num (X,Y,Z), index range lo/hi)
        "tiny": (1, (2, 3, 4), 5, (6, 5, 1), (-2, 8)),
        "b2_c80": (2, (3, 4, 5), 80, (8, 7, 1), (-3, 11)),
        "z2": (1, (4, 4, 4), 16, (5, 5, 2), (-1, 6)),
        "dups": (2, (5, 6, 7), 12, (3, 2, 1), (0, 3)),     # heavy collisions
        "all_out": (1, (2, 2, 2), 8, (4, 4, 1), (10, 20)),  # nothing lands
    }
    for name, (B, (D, H, W), C, (X, Y, Z), (lo, hi)) in cases.items():
        geom = torch.randint(lo, hi, (B, 1, D, H, W, 3), generator=gen, dtype=torch.int32)
        # integer-valued floats: every summation order gives the same bits
        feats = torch.randint(-8, 9, (B, 1, D, H, W, C), generator=gen).float().requires_grad_(True)
        voxel_num = torch.tensor([X, Y, Z])
        outp = VP.voxel_pooling(geom, feats, voxel_num)                 # [B, C, Y, X] view
        gout = torch.randint(-4, 5, outp.shape, generator=gen).float()
        outp.backward(gout)
        vp[f"{name}/geom_xyz"] = geom.numpy()
        vp[f"{name}/feats"] = feats.detach().numpy()
        vp[f"{name}/voxel_num"] = voxel_num.numpy()
        vp[f"{name}/out"] = outp.detach().contiguous().numpy()
        vp[f"{name}/grad_out"] = gout.numpy()
        vp[f"{name}/grad_feats"] = feats.grad.numpy()
    # one case with real-valued features (tolerance test) on the "tiny" geometry
    geom = torch.from_numpy(vp["b2_c80/geom_xyz"])
    feats = torch.randn(2, 1, 3, 4, 5, 80, generator=gen)
    vp["b2_c80_randn/geom_xyz"] = geom.numpy()
    vp["b2_c80_randn/feats"] = feats.numpy()
    vp["b2_c80_randn/voxel_num"] = vp["b2_c80/voxel_num"]
    vp["b2_c80_randn/out"] = VP.voxel_pooling(geom, feats, torch.tensor([8, 7, 1])).contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "voxel_pooling.npz"), **vp)

    # ---------------- B3: lift = softmax over D (x) context, lss_fpn.py:462-466,486 --------------
    lift = {}
    B, D, C, fH, fW = 2, 6, 8, 3, 4
    height_feature = torch.randn(B, D + C, fH, fW, generator=gen)
    height = height_feature[:, :D].softmax(1)                                               # :462
    prod = height.unsqueeze(1) * height_feature[:, D:(D + C)].unsqueeze(2)                  # :464-466
    prod = prod.reshape(B, 1, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2).contiguous()           # :469-486
    lift["height_feature"] = height_feature.numpy()
    lift["lifted"] = prod.numpy()
    lift["dims"] = np.array([B, D, C, fH, fW])
    np.savez_compressed(os.path.join(OUT, "lift.npz"), **lift)
    for fn in ("frustum.npz", "geometry.npz", "voxel_pooling.npz", "lift.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


# ------------------------------------------------------------------ input contract (dataset helpers)
def make_input_contract():
    """Outputs of the reference's dataset-side geometry helpers (dataset/nusc_mv_det_dataset.py:41-86,
    130-161 img_transform's matrix, 164-179 bev_transform's matrix, 433-446 sample_ida_augmentation)."""
    from scipy.spatial.transform import Rotation

    class _Img:                                   # the only PIL.Image methods img_transform touches
        def resize(self, *_a, **_k): return self
        def crop(self, *_a, **_k): return self
        def transpose(self, *_a, **_k): return self
        def rotate(self, *_a, **_k): return self

    _stub('cv2', Rodrigues=lambda v: (Rotation.from_rotvec(np.asarray(v, np.float64)).as_matrix(), None))
    for name in ('mmcv', 'mmdet3d', 'mmdet3d.core', 'imageio', 'skimage', 'nuscenes', 'nuscenes.utils', 'pyquaternion', 'PIL', 'mmdet3d.core.bbox',
                 'mmdet3d.core.bbox.structures', 'dataset.transforms'):
        _stub(name)
    _stub('skimage.transform', rotate=_raise, warp=_raise, resize=_raise)
    _stub('mmdet3d.core.bbox.structures.lidar_box3d', LiDARInstance3DBoxes=object)
    _stub('nuscenes.utils.data_classes', Box=object)
    sys.modules['PIL'].Image = types.SimpleNamespace(ANTIALIAS=0, BICUBIC=0, FLIP_LEFT_RIGHT=0)
    sys.modules['pyquaternion'].Quaternion = object
    sys.modules['dataset.transforms'].ResizeLongestSide = object
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    ds = importlib.import_module('dataset.nusc_mv_det_dataset')
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from sgv3d_amd import synthetic as S
    out = {}
    poses = [(11.0, 5.5, 0.0, 0.0), (5.0, 4.0, 2.0, 0.5), (20.0, 8.0, -3.0, -1.0), (14.5, 6.3, 7.0, 2.0)]
    s2e = np.stack([S.make_calib(pitch_deg=p, cam_h=hh, yaw_deg=y, roll_deg=r)['sensor2ego'] for p, hh, y, r in poses])
    out['sensor2ego'] = s2e
    den, s2v, ref = [], [], []
    for m in s2e:
        e2s = np.linalg.inv(m.astype(np.float64))
        d = ds.get_denorm(e2s)
        den.append(d)
        s2v.append(ds.get_sensor2virtual(d))
        ref.append(ds.get_reference_height(d))
    out['denorm'], out['sensor2virtual'], out['reference_height'] = np.stack(den), np.stack(s2v), np.stack(ref)
    pts = np.array([[0.3, -1.2, 4.0], [2.0, 0.5, 3.5], [-1.0, 0.7, 6.0]])
    out['plane_points'], out['plane'] = pts, ds.equation_plane(pts)
    # ida: eval-mode sampling for the DAIR (1080x1920 -> 864x1536) and a letterboxed case, + its matrix
    fake = types.SimpleNamespace(ida_aug_conf={'H': 1080, 'W': 1920, 'final_dim': (864, 1536), 'bot_pct_lim': (0.0, 0.0)})
    r = ds.NuscMVDetDataset.sample_ida_augmentation(fake)
    out['ida_dair_sample'] = np.array([r[0], *r[1], *r[2], float(r[3]), float(r[4])], dtype=np.float64)
    out['ida_dair_mat'] = ds.img_transform(_Img(), *r)[1].numpy()
    fake.ida_aug_conf = {'H': 900, 'W': 1600, 'final_dim': (256, 704), 'bot_pct_lim': (0.0, 0.22)}
    r = ds.NuscMVDetDataset.sample_ida_augmentation(fake)
    out['ida_nusc_sample'] = np.array([r[0], *r[1], *r[2], float(r[3]), float(r[4])], dtype=np.float64)
    out['ida_nusc_mat'] = ds.img_transform(_Img(), *r)[1].numpy()
    out['ida_aug_args'] = np.array([0.47, 12.0, 30.0, 12.0 + 704, 30.0 + 256, 1.0, 5.4])
    out['ida_aug_mat'] = ds.img_transform(_Img(), 0.47, (752, 423), (12, 30, 12 + 704, 30 + 256), True, 5.4)[1].numpy()
    # bda
    out['bda_identity'] = ds.bev_transform(torch.zeros(0, 9), 0, 1.0, False, False)[1].numpy()
    out['bda_aug_args'] = np.array([22.5, 1.05, 1.0, 0.0])
    out['bda_aug'] = ds.bev_transform(torch.zeros(0, 9), 22.5, 1.05, True, False)[1].numpy()
    np.savez_compressed(os.path.join(OUT, "input_contract.npz"), **out)
    print("input_contract.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if "--input-contract" not in sys.argv:      # `--input-contract`: only (re)generate input_contract.npz
        main()
    make_input_contract()
