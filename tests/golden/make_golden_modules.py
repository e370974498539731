#!/usr/bin/env python3
"""Golden vectors for the reference's OWN torch modules on the hot path (tests/golden/modules.npz),
produced by EXECUTING THE REFERENCE'S PYTHON in the build container (needs /root/reference; never runs
on the GPU box).  Complements make_golden.py (geometry / voxel pooling / lift).

Executed unmodified from the reference, in eval mode, on seeded random weights and inputs:

* ``ASPP``, ``Mlp``, ``SELayer``                       layers/backbones/lss_fpn.py:49-159
* ``HeightNet.forward`` (incl. the 27-vector)          lss_fpn.py:207-250
* ``LSSFPN._forward_single_sweep``                     lss_fpn.py:422-495  (neck features -> BEV map)
* ``SABlock``, ``TaskHead``, ``TaskFPN``               layers/backbones/bsm_lss_fpn.py:151-212
* ``MSCThead.forward``                                 bsm_lss_fpn.py:259-320
* ``BSMLSSFPN._forward_single_sweep``                  bsm_lss_fpn.py:485-559 (softmax / concat / 0.45 mask / lift / pool)

What is NOT the reference's: the third-party blocks those modules instantiate (mmdet ``BasicBlock``,
mmcv ``DCN``) are absent from this image and are supplied by the small restatements below (the DCN
one is built on ``F.grid_sample``, independently of oracle/torch_model.py); image backbone + neck are
replaced by random feature maps (``get_cam_feats`` is overridden); ``Tensor.cuda`` is the identity
(lss_fpn.py:491 calls ``self.voxel_num.cuda()``) and the CUDA extension is the literal stub of
make_golden.py.  So these vectors pin the reference-owned arithmetic and composition; the third-party
classes stay "parity unpinned" (DESIGN.md §4).

Outputs are DATA ONLY (weights, inputs, expected outputs).

    python tests/golden/make_golden_modules.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stub helpers, calibration builder)

REF = MG.REF


# ------------------------------------------------------------------ third-party block restatements
class BasicBlock(nn.Module):
    """mmdet 2.19.0 ``BasicBlock(inplanes, planes)`` at stride 1 without downsample (the only form
    lss_fpn.py:186-188 / bsm_lss_fpn.py:185-186 use): conv3x3-BN-ReLU-conv3x3-BN, + identity, ReLU."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + x)


class DCN(nn.Module):
    """mmcv-full 1.4.0 ``DeformConv2dPack`` (3x3, stride 1, pad 1, deform_groups 1, no bias): offsets
    from ``conv_offset`` (channel 2t = dy, 2t+1 = dx of tap t, row-major), bilinear sampling with
    zeros outside the image (``grid_sample(padding_mode='zeros', align_corners=True)`` on pixel
    coordinates), grouped 3x3 weights."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, groups=1, **_unused):
        super().__init__()
        assert kernel_size == 3 and padding == 1
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, 3, 3))
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')
        self.conv_offset = nn.Conv2d(in_channels, 18, 3, 1, 1, bias=True)

    def forward(self, x):
        B, C, H, W = x.shape
        off = self.conv_offset(x)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
        cols = []
        for t in range(9):
            hf = ys[None] - 1 + t // 3 + off[:, 2 * t]
            wf = xs[None] - 1 + t % 3 + off[:, 2 * t + 1]
            grid = torch.stack([2 * wf / (W - 1) - 1, 2 * hf / (H - 1) - 1], -1)
            cols.append(F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True))
        col = torch.stack(cols, 2)                                      # [B, C, 9, H, W]
        g = self.groups
        cpg, opg = C // g, self.weight.shape[0] // g
        outs = []
        for gi in range(g):
            wg = self.weight[gi * opg:(gi + 1) * opg].reshape(opg, cpg * 9)
            cg = col[:, gi * cpg:(gi + 1) * cpg].reshape(B, cpg * 9, H * W)
            outs.append((wg @ cg).reshape(B, opg, H, W))
        return torch.cat(outs, 1)


def _build_conv_layer(cfg, *a, **k):
    cfg = dict(cfg)
    assert cfg.pop('type') == 'DCN'
    return DCN(**cfg)


def import_reference():
    S = MG._stub
    S('mmcv')
    S('mmcv.cnn', build_conv_layer=_build_conv_layer)
    S('mmdet')
    S('mmdet.models', build_backbone=MG._raise)
    S('mmdet.models.backbones')
    S('mmdet.models.backbones.resnet', BasicBlock=BasicBlock)
    S('mmdet.core', reduce_mean=MG._raise)
    S('mmdet3d')
    S('mmdet3d.models', build_neck=MG._raise)
    S('mmdet3d.core', draw_heatmap_gaussian=MG._raise, gaussian_radius=MG._raise)
    S('mmdet3d.models.dense_heads')
    S('mmdet3d.models.dense_heads.centerpoint_head', CenterHead=MG._Dummy)
    S('mmdet3d.models.utils', clip_sigmoid=MG._raise)
    S('cv2')
    sys.path.insert(0, REF)
    ext = types.ModuleType('ops.voxel_pooling.voxel_pooling_ext')
    ext.voxel_pooling_forward_wrapper = MG._kernel_stub
    sys.modules['ops.voxel_pooling.voxel_pooling_ext'] = ext
    import layers.backbones.lss_fpn as L
    import layers.backbones.bsm_lss_fpn as Bm
    torch.Tensor.cuda = lambda self, *a, **k: self       # lss_fpn.py:491 / bsm_lss_fpn.py:555 on a CPU build
    return L, Bm


# ------------------------------------------------------------------ helpers
def randomize_(m, gen):
    """Seeded non-trivial BatchNorm statistics / affine, DCN offsets and O(1) logits."""
    with torch.no_grad():
        for name, mod in m.named_modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                n = mod.num_features
                mod.weight.copy_(1.0 + 0.2 * torch.randn(n, generator=gen))
                mod.bias.copy_(0.2 * torch.randn(n, generator=gen))
                mod.running_mean.copy_(0.2 * torch.randn(n, generator=gen))
                mod.running_var.copy_(0.7 + 0.6 * torch.rand(n, generator=gen))
            if name.endswith('conv_offset'):
                mod.weight.copy_(0.05 * torch.randn(mod.weight.shape, generator=gen))
                mod.bias.copy_(0.7 * torch.randn(mod.bias.shape, generator=gen))
            if name.endswith(('height_layer', 'context_conv', 'depth_head1.head', 'context_conv1.3')):
                mod.weight.mul_(6.0)
            if name.endswith('semantic_head1.head'):
                mod.weight.mul_(60.0)
                mod.bias[0] += 1.2
    return m.eval()


def put(out, tag, module=None, **arrays):
    if module is not None:
        for k, v in module.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.detach().numpy().copy()
    for k, v in arrays.items():
        out[f"{tag}/{k}"] = v.detach().numpy().copy() if isinstance(v, torch.Tensor) else np.asarray(v)


def make_mats(batch, scale, num_cams=1):
    """mats_dict [B, 1, num_cams, 4, 4] from make_golden.make_calib (float32 matrices)."""
    cams = []
    for i in range(batch * num_cams):
        cams.append(MG.make_calib(pitch_deg=11.0 + 2.5 * i, cam_h=5.5 + 0.5 * i, yaw_deg=1.5 * i, roll_deg=0.4 * i,
                                  fx=2183.375 * scale, fy=2329.2976 * scale, cx=940.59 * scale, cy=567.568 * scale,
                                  crop=(1.5 * i, 0.5 * i)))
    t = lambda k: torch.from_numpy(np.stack([c[k] for c in cams])).view(batch, 1, num_cams, 4, 4)
    bda = torch.eye(4).repeat(batch, 1, 1)
    for b in range(batch):       # a mild BEV augmentation matrix (rotation about z + scale), exercised by the 27-vector
        a = 0.03 * (b + 1)
        bda[b, 0, 0], bda[b, 0, 1], bda[b, 1, 0], bda[b, 1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        bda[b, :3, :3] *= 1.0 + 0.01 * b
    return {
        'sensor2ego_mats': t('sensor2ego'), 'intrin_mats': t('intrin'), 'ida_mats': t('ida'),
        'sensor2sensor_mats': torch.eye(4).view(1, 1, 1, 4, 4).repeat(batch, 1, num_cams, 1, 1),
        'sensor2virtual_mats': t('sensor2virtual'),
        'reference_heights': torch.tensor([float(c['reference_height']) for c in cams]).view(batch, 1, num_cams),
        'bda_mat': bda,
    }


def bare_lss(cls, final_dim, downsample, d_bound, xb, yb, zb, out_channels):
    """An (BSM)LSSFPN instance with the geometry buffers of its __init__ (lss_fpn.py:275-293 /
    bsm_lss_fpn.py:343-361) but without the third-party backbone / neck."""
    obj = cls.__new__(cls)
    nn.Module.__init__(obj)
    obj.downsample_factor = downsample
    obj.d_bound = d_bound
    obj.final_dim = final_dim
    obj.output_channels = out_channels
    obj.is_train_height = False
    rows = [xb, yb, zb]
    obj.register_buffer('voxel_size', torch.Tensor([r[2] for r in rows]))
    obj.register_buffer('voxel_coord', torch.Tensor([r[0] + r[2] / 2.0 for r in rows]))
    obj.register_buffer('voxel_num', torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows]))
    obj.register_buffer('frustum', obj.create_frustum())
    obj.height_channels = obj.frustum.shape[0]
    return obj


def main(out_dir=None):
    L, Bm = import_reference()
    gen = torch.Generator().manual_seed(20261003)
    rn = lambda *s: torch.randn(*s, generator=gen)
    out = {}
    site = [0]

    def seeded(ctor, *a, **k):
        """Constructors draw their initial weights from the GLOBAL generator: reseed it per construction site, so the
        fixture regenerates bit for bit (tests/test_golden_regen_cpu.py)."""
        site[0] += 1
        torch.manual_seed(20261007 + 1000 * site[0])
        return ctor(*a, **k)
    with torch.no_grad():
        # ---------------- ASPP / Mlp / SELayer  (lss_fpn.py:49-159) -------------------------------
        m = randomize_(seeded(L.ASPP, 16, 16), gen)
        x = rn(2, 16, 9, 11)
        put(out, "aspp", m, x=x, y=m(x))
        m = seeded(L.Mlp, 27, 16, 12).eval()
        x = rn(3, 27)
        put(out, "mlp", m, x=x, y=m(x))
        m = seeded(L.SELayer, 16).eval()
        x, xs = rn(2, 16, 5, 7), rn(2, 16, 1, 1)
        put(out, "se", m, x=x, x_se=xs, y=m(x, xs))

        # ---------------- HeightNet.forward  (lss_fpn.py:207-250), two cameras per sample ----------
        hn = randomize_(seeded(L.HeightNet, 24, 16, 8, 6), gen)
        mats = make_mats(2, 112 / 1536, num_cams=2)
        seen = {}
        h = hn.bn.register_forward_pre_hook(lambda mod, inp: seen.__setitem__('v', inp[0].clone()))
        x = rn(4, 24, 5, 7)
        y = hn(x, mats)
        h.remove()
        put(out, "heightnet", hn, x=x, y=y, mlp_input=seen['v'], **{"mats/" + k: v for k, v in mats.items()})
        assert seen['v'].shape == (4, 27)

        # ---------------- LSSFPN._forward_single_sweep  (lss_fpn.py:422-495) ------------------------
        xb, yb, zb = [0, 25.6, 0.4], [-12.8, 12.8, 0.4], [-5, 3, 8]
        lss = bare_lss(L.LSSFPN, (80, 112), 16, [-2.0, 0.0, 6], xb, yb, zb, 8)
        lss.height_net = hn
        lss.assist_layer = seeded(nn.Conv2d, 24, 4, 1)
        mats = make_mats(2, 112 / 1536)
        feats = rn(2, 1, 1, 24, 5, 7)
        lss.get_cam_feats = lambda imgs: feats
        lss.eval()
        bev = lss._forward_single_sweep(0, torch.zeros(2, 1, 1, 3, 80, 112), mats)
        geom = lss.get_geometry(mats['sensor2ego_mats'][:, 0], mats['sensor2virtual_mats'][:, 0], mats['intrin_mats'][:, 0],
                                mats['ida_mats'][:, 0], mats['reference_heights'][:, 0], mats['bda_mat'])
        q = (geom - (lss.voxel_coord - lss.voxel_size / 2.0)) / lss.voxel_size
        assert torch.isfinite(q).all() and q.abs().max() < 2e9      # CPU .int() == GPU cast on this fixture
        put(out, "lss_sweep", hn, feats=feats, bev=bev, geom_xyz=q.int(), frustum=lss.frustum,
            bounds=np.array([xb, yb, zb], np.float64), cfg=np.array([80, 112, 16, -2.0, 0.0, 6], np.float64),
            **{"mats/" + k: v for k, v in mats.items()})
        gi, vn = q.int(), lss.voxel_num
        inr = ((gi[..., 0] >= 0) & (gi[..., 0] < vn[0]) & (gi[..., 1] >= 0) & (gi[..., 1] < vn[1])
               & (gi[..., 2] >= 0) & (gi[..., 2] < vn[2])).float().mean()
        print(f"lss_sweep: bev {tuple(bev.shape)} points in range {inr:.3f}")
        assert 0.3 < inr < 1.0, "fixture must have points inside and outside the grid"

        # ---------------- SABlock / TaskHead / TaskFPN  (bsm_lss_fpn.py:151-212) -------------------
        m = seeded(Bm.SABlock, 8, 8).eval()
        x, y = rn(2, 8, 6, 5), rn(2, 8, 6, 5)
        put(out, "sablock", m, x=x, y_in=y, y=m(x, y))
        m = randomize_(seeded(Bm.TaskHead, 8, 8, 5), gen)
        x = rn(2, 8, 6, 5)
        logits, feat = m(x)
        put(out, "taskhead", m, x=x, logits=logits, feat=feat, logits_only=m(x, return_feat=False))
        m = seeded(Bm.TaskFPN, 8, 6).eval()
        f0, f1 = rn(2, 8, 5, 7), rn(2, 6, 10, 14)
        put(out, "taskfpn", m, feat0=f0, feat1=f1, y=m(f0, f1))

        # ---------------- MSCThead.forward  (bsm_lss_fpn.py:259-320) --------------------------------
        ms = randomize_(seeded(Bm.MSCThead, in_channels=[24, 20], mid_channels=[16, 12], depth_channels=10,
                                   semantic_channels=7, context_channels=8), gen)
        mats = make_mats(2, 112 / 1536)
        x0, x1 = rn(2, 1, 24, 5, 7), rn(2, 1, 20, 10, 14)
        d1, s1, c1, s0 = ms([x0, x1], mats)
        put(out, "mscthead", ms, x0=x0, x1=x1, depth1=d1, semantic1=s1, context1=c1, semantic0=s0,
            **{"mats/" + k: v for k, v in mats.items()})

        # ---------------- BSMLSSFPN._forward_single_sweep  (bsm_lss_fpn.py:485-559) -----------------
        bsm = bare_lss(Bm.BSMLSSFPN, (80, 112), 16 // 2, [-2.0, 3.5, 10], xb, yb, zb, 8)     # :343 halves the factor
        bsm.height_net = ms
        bsm.get_cam_feats = lambda imgs: [x0, x1]
        bsm.is_train_height = True
        bsm.eval()
        bev, (sem0, sem1) = bsm._forward_single_sweep(0, torch.zeros(2, 1, 1, 3, 80, 112), mats)
        semantic = s1.softmax(dim=1)
        frac_bg = (semantic[:, 0] > 0.45).float().mean()
        print(f"bsm_sweep: bev {tuple(bev.shape)} background-masked pixels {frac_bg:.3f}")
        assert 0.1 < frac_bg < 0.9, "fixture must exercise both sides of the 0.45 mask"
        assert torch.equal(sem0, s0) and torch.equal(sem1, s1)
        put(out, "bsm_sweep", None, bev=bev, frustum=bsm.frustum, bounds=np.array([xb, yb, zb], np.float64),
            cfg=np.array([80, 112, 8, -2.0, 3.5, 10], np.float64))
    path = os.path.join(out_dir or os.environ.get("SGV3D_GOLDEN_OUT", HERE), "modules.npz")
    np.savez_compressed(path, **out)
    print("modules.npz", os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main(sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
