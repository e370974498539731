"""Configs and synthetic inputs for benchmarks, smoke runs and tests (no dataset, no checkpoint).

* ``r50_256_conf()`` etc. reproduce the config dicts of the reference's experiment files verbatim
  (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:33-172) so the same dicts drive
  this build and, in the reference harness, the reference model.
* ``make_mats`` builds a DAIR-V2X-like ``mats_dict`` with the dataset's formulas
  (dataset/nusc_mv_det_dataset.py:63-86 get_denorm / get_sensor2virtual / get_reference_height,
  :433-446 sample_ida_augmentation, :864-871 collate layout) — SURVEY.md §8(d) "Synthetic inputs".
* ``randomize_norm_stats_`` perturbs BatchNorm affine/running statistics so BN folding is exercised
  (fresh BN layers are the identity, and mmdet zero-initialises the last BN of every residual block).
"""
import copy
import math

import numpy as np
import torch

H, W = 1080, 1920
final_dim = (864, 1536)

_backbone_conf = {
    'x_bound': [0, 102.4, 0.4],
    'y_bound': [-51.2, 51.2, 0.4],
    'z_bound': [-5, 3, 8],
    'd_bound': [-2.0, 0.0, 90],
    'final_dim': final_dim,
    'output_channels': 80,
    'downsample_factor': 16,
    'img_backbone_conf': dict(
        type='ResNet',
        depth=50,
        frozen_stages=0,
        out_indices=[0, 1, 2, 3],
        norm_eval=False,
        init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50'),
    ),
    'img_neck_conf': dict(
        type='SECONDFPN',
        in_channels=[256, 512, 1024, 2048],
        upsample_strides=[0.25, 0.5, 1, 2],
        out_channels=[128, 128, 128, 128],
    ),
    'height_net_conf': dict(in_channels=512, mid_channels=512),
    'is_train_height': False,
    'is_bsm': False,
}

_bev_backbone = dict(type='ResNet', in_channels=80, depth=18, num_stages=3, strides=(1, 2, 2), dilations=(1, 1, 1),
                     out_indices=[0, 1, 2], norm_eval=False, base_channels=160)
_bev_neck = dict(type='SECONDFPN', in_channels=[80, 160, 320, 640], upsample_strides=[1, 2, 4, 8],
                 out_channels=[64, 64, 64, 64])

CLASSES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle',
           'pedestrian', 'traffic_cone']
TASKS = [
    dict(num_class=1, class_names=['car']),
    dict(num_class=2, class_names=['truck', 'construction_vehicle']),
    dict(num_class=2, class_names=['bus', 'trailer']),
    dict(num_class=1, class_names=['barrier']),
    dict(num_class=2, class_names=['motorcycle', 'bicycle']),
    dict(num_class=2, class_names=['pedestrian', 'traffic_cone']),
]
common_heads = dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))
bbox_coder = dict(type='CenterPointBBoxCoder', post_center_range=[0.0, -61.2, -10.0, 122.4, 61.2, 10.0], max_num=500,
                  score_threshold=0.1, out_size_factor=4, voxel_size=[0.1, 0.1, 8], pc_range=[0, -51.2, -5, 104.4, 51.2, 3],
                  code_size=9)
train_cfg = dict(point_cloud_range=[0, -51.2, -5, 102.4, 51.2, 3], grid_size=[1024, 1024, 1], voxel_size=[0.1, 0.1, 8],
                 out_size_factor=4, dense_reg=1, gaussian_overlap=0.1, max_objs=500, min_radius=2,
                 code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5])
test_cfg = dict(post_center_limit_range=[0.0, -61.2, -10.0, 122.4, 61.2, 10.0], max_per_img=500, max_pool_nms=False,
                min_radius=[4, 12, 10, 1, 0.85, 0.175], score_threshold=0.1, out_size_factor=4, voxel_size=[0.1, 0.1, 8],
                nms_type='circle', pre_max_size=1000, post_max_size=83, nms_thr=0.2)
_head_conf = {
    'bev_backbone_conf': _bev_backbone,
    'bev_neck_conf': _bev_neck,
    'tasks': TASKS,
    'common_heads': common_heads,
    'bbox_coder': bbox_coder,
    'train_cfg': train_cfg,
    'test_cfg': test_cfg,
    'in_channels': 256,  # Equal to bev_neck output_channels.
    'loss_cls': dict(type='GaussianFocalLoss', reduction='mean'),
    'loss_bbox': dict(type='L1Loss', reduction='mean', loss_weight=0.25),
    'gaussian_overlap': 0.1,
    'min_radius': 2,
}


def r50_256_conf():
    """BASELINE cfg-2: R50, 864x1536 -> 256x256 BEV (the reference's DAIR-V2X experiment file)."""
    return copy.deepcopy(_backbone_conf), copy.deepcopy(_head_conf)


def r101_256_conf():
    b, h = r50_256_conf()
    b['img_backbone_conf']['depth'] = 101
    return b, h


def r101_512_conf():
    """BASELINE cfg-3 geometry: R101, 1080x1920 images padded to 1088x1920 (a multiple of the stride-16
    feature grid), 512x512 BEV grid (0.2 m cells).  This build computes it in fp32."""
    b, h = r101_256_conf()
    b['final_dim'] = (1088, 1920)
    b['x_bound'] = [0, 102.4, 0.2]
    b['y_bound'] = [-51.2, 51.2, 0.2]
    h['bbox_coder'] = dict(h['bbox_coder'], out_size_factor=4, voxel_size=[0.05, 0.05, 8])
    h['test_cfg'] = dict(h['test_cfg'], voxel_size=[0.05, 0.05, 8])
    h['train_cfg'] = dict(h['train_cfg'], grid_size=[2048, 2048, 1], voxel_size=[0.05, 0.05, 8])
    return b, h


def bsm_r101_256_conf():
    """BASELINE cfg-5 model: SGV3D BSM, R101, 864x1536 -> 256x256 BEV
    (exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:42-101)."""
    b, h = r50_256_conf()
    b['d_bound'] = [-2.0, 3.5, 180]
    b['img_backbone_conf']['depth'] = 101
    b['img_backbone_conf']['init_cfg'] = dict(type='Pretrained', checkpoint='torchvision://resnet101')
    b['height_net_conf'] = dict(in_channels=[512, 512], mid_channels=[512, 256], semantic_channels=7)
    b['is_bsm'] = True
    h['bev_backbone_conf'] = dict(type='ResNet', in_channels=87, depth=18, num_stages=3, strides=(1, 2, 2),
                                  dilations=(1, 1, 1), out_indices=[0, 1, 2], norm_eval=False, base_channels=174)
    h['bev_neck_conf'] = dict(type='SECONDFPN', in_channels=[87, 174, 348, 696], upsample_strides=[1, 2, 4, 8],
                              out_channels=[64, 64, 64, 64])
    return b, h


def small_bsm_conf(final=(128, 192), bev=64, depth=18):
    """Reduced-size BSM config with the cfg-5 layer structure (tests)."""
    b, h = bsm_r101_256_conf()
    sb, _ = small_conf(final, bev, depth)
    for k in ('final_dim', 'x_bound', 'y_bound'):
        b[k] = sb[k]
    b['d_bound'] = [-2.0, 3.5, 12]
    b['img_backbone_conf']['depth'] = depth
    if depth in (18, 34):
        b['img_neck_conf']['in_channels'] = [64, 128, 256, 512]
    return b, h


def small_conf(final=(128, 192), bev=64, depth=18):
    """Reduced geometry with the same layer structure (tests: the CPU oracle finishes in seconds)."""
    b, h = r50_256_conf()
    b['final_dim'] = final
    b['d_bound'] = [-2.0, 0.0, 12]
    half = bev * 0.4 / 2
    b['x_bound'] = [0, bev * 0.4, 0.4]
    b['y_bound'] = [-half, half, 0.4]
    if depth != 50:
        b['img_backbone_conf']['depth'] = depth
        if depth in (18, 34):
            b['img_neck_conf']['in_channels'] = [64, 128, 256, 512]
    return b, h


# -------------------------------------------------------------------------------------------------
from .input_contract import get_denorm, get_reference_height, get_sensor2virtual, rodrigues as _rodrigues


def make_calib(pitch_deg=11.0, cam_h=5.5, yaw_deg=0.0, roll_deg=0.0, fx=2183.375, fy=2329.2976, cx=940.59,
               cy=567.568, resize=0.8, crop=(0.0, 0.0)):
    """One roadside camera: (x right, y down, z forward) looking along ego +x, pitched down."""
    f32 = np.float32
    p, yw, rl = (math.radians(a) for a in (pitch_deg, yaw_deg, roll_deg))
    fwd = np.array([math.cos(p), 0.0, -math.sin(p)])
    right = np.array([0.0, -1.0, 0.0])
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    Rroll = _rodrigues(np.array([0.0, 0.0, 1.0]) * rl)
    Rz = np.array([[math.cos(yw), -math.sin(yw), 0], [math.sin(yw), math.cos(yw), 0], [0, 0, 1]])
    s2e = np.eye(4)
    s2e[:3, :3] = Rz @ R @ Rroll
    s2e[:3, 3] = [0.0, 0.0, cam_h]
    e2s = np.linalg.inv(s2e)
    denorm = get_denorm(e2s)                                                 # dataset/nusc_mv_det_dataset.py:63-86
    s2v = get_sensor2virtual(denorm)
    refh = get_reference_height(denorm)
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    ida = np.eye(4)
    ida[0, 0] = ida[1, 1] = resize
    ida[0, 3], ida[1, 3] = -crop[0], -crop[1]
    return dict(sensor2ego=s2e.astype(f32), sensor2virtual=s2v.astype(f32), intrin=K.astype(f32),
                ida=ida.astype(f32), bda=np.eye(4, dtype=f32), reference_height=refh)


def make_mats(batch, device='cpu', scale=1.0, vary=True):
    """mats_dict for ``batch`` samples x 1 sweep x 1 camera (collate layout dataset/...:864-871).
    ``scale`` shrinks the intrinsics for reduced-resolution test configs."""
    cams = []
    for b in range(batch):
        kw = dict(fx=2183.375 * scale, fy=2329.2976 * scale, cx=940.59 * scale, cy=567.568 * scale)
        if vary and b > 0:
            kw.update(pitch_deg=11.0 + 1.5 * b, cam_h=5.5 + 0.4 * b, yaw_deg=1.0 * b, roll_deg=0.3 * b)
        cams.append(make_calib(**kw))
    t = lambda k: torch.from_numpy(np.stack([c[k] for c in cams])).view(batch, 1, 1, 4, 4).to(device)
    return {
        'sensor2ego_mats': t('sensor2ego'),
        'intrin_mats': t('intrin'),
        'ida_mats': t('ida'),
        'sensor2sensor_mats': torch.eye(4).view(1, 1, 1, 4, 4).repeat(batch, 1, 1, 1, 1).to(device),
        'sensor2virtual_mats': t('sensor2virtual'),
        'reference_heights': torch.tensor([float(c['reference_height']) for c in cams]).view(batch, 1, 1).to(device),
        'bda_mat': torch.from_numpy(np.stack([c['bda'] for c in cams])).to(device),
    }


def make_images(batch, final=final_dim, device='cpu', seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 1, 1, 3, final[0], final[1], generator=g).to(device)


def randomize_norm_stats_(model, seed=0, dcn_offsets=True, residual_gamma=None):
    """Seeded perturbation of every BatchNorm (weight, bias, running stats) and of the DCN offset
    conv (zero-initialised => plain conv), so folded-BN and deformable sampling are really tested.

    ``residual_gamma`` (e.g. 0.3): scale of the last BatchNorm of every residual block.  mmdet zero-initialises it
    (``zero_init_residual``) and training keeps it small; with gamma ~ 1 and untrained running statistics every block
    doubles the variance of its input (16 bottlenecks: activations ~ 1e2 at the image neck, ~ 2e2 in the BEV map), which
    turns an absolute 1e-3 parity bar into a 1e-5 relative one -- below the rounding noise of ANY float32 execution
    (tools/parity_scale_probe.py measures both float32 paths against a float64 evaluation)."""
    from .layers import blocks as _blocks
    g = torch.Generator().manual_seed(seed)
    last_bn = set()
    if residual_gamma is not None:
        for m in model.modules():
            if isinstance(m, _blocks.Bottleneck):
                last_bn.add(m.bn3)
            elif isinstance(m, _blocks.BasicBlock):
                last_bn.add(m.bn2)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                n = m.num_features
                m.weight.copy_(1.0 + 0.1 * torch.randn(n, generator=g))
                m.bias.copy_(0.1 * torch.randn(n, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                m.running_var.copy_(1.0 + 0.1 * torch.rand(n, generator=g))
                if m in last_bn:
                    m.weight.mul_(residual_gamma)
            if dcn_offsets and name.endswith('conv_offset'):
                m.weight.copy_(0.02 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.5 * torch.randn(m.bias.shape, generator=g))
            # default-initialised 1x1 heads give near-zero logits / context: scale them to O(1) so the
            # softmax, the BEV map and the head see realistic dynamic range
            if name.endswith(('height_layer', 'context_conv', 'depth_head1.head', 'context_conv1.3')):
                m.weight.mul_(8.0)
            # BSM: make the background class win on part of the image so the 0.45 mask is exercised
            if name.endswith('semantic_head1.head'):
                m.weight.mul_(20.0)
                m.bias[0] += 0.5
    return model


def calibrate_norm_stats_(model, imgs, mats):
    """Running statistics of the BatchNorm2d layers := the batch statistics of ``imgs`` (one training-mode forward of
    the HIP path with momentum 1), i.e. what training leaves behind: every normalised activation is O(1), as in a
    trained network.  Random-initialised weights with running statistics (0, 1) let the activations drift with depth
    (the BEV map of a near-camera voxel sums ~250 rows, the head trunk adds seven residual stages on top: |outputs| ~
    1e2), which turns the absolute 1e-3 parity bar into a 1e-5 relative one -- fp32 rounding noise of the reference
    itself.  Layers whose statistics would come from fewer than 64 values per channel (the 27-feature BatchNorm1d, the
    ASPP image-pooling branch) keep their statistics.  GPU only (the training forward has no CPU path)."""
    bns = [(n, m) for n, m in model.named_modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d))]
    keep = {n: (m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for n, m in bns
            if isinstance(m, torch.nn.BatchNorm1d) or 'global_avg_pool' in n}
    mom = {n: m.momentum for n, m in bns}
    drops = [(m, m.p) for m in model.modules() if isinstance(m, torch.nn.Dropout)]
    was_training = model.training
    try:
        for n, m in bns:
            m.momentum = 1.0
        for m, _ in drops:
            m.p = 0.0
        model.train()
        with torch.no_grad():
            model(imgs, mats)
        torch.cuda.synchronize()
    finally:
        for n, m in bns:
            m.momentum = mom[n]
        for m, p in drops:
            m.p = p
        model.train(was_training)
    with torch.no_grad():
        for n, m in bns:
            if n in keep:
                m.running_mean.copy_(keep[n][0]); m.running_var.copy_(keep[n][1]); m.num_batches_tracked.copy_(keep[n][2])
            else:
                m.running_var.clamp_(min=1e-3)
    return model


def make_gt(batch, seed=0, n_range=(0, 40), num_labels=10, pc_range=(0, -51.2, 102.4, 51.2), stress=True):
    """Seeded ground-truth boxes [N, 9] (x, y, z, w, l, h, yaw, vx, vy) and labels [N] per sample, CPU tensors.
    With ``stress`` a few boxes are placed outside the range, just below the lower edges (cell truncation) and
    with a non-positive size, which the target assignment has to skip without losing their slot."""
    rng = np.random.default_rng(seed)
    boxes, labels = [], []
    for b in range(batch):
        n = int(rng.integers(n_range[0], n_range[1] + 1))
        bx = np.zeros((n, 9), np.float32)
        bx[:, 0] = rng.uniform(pc_range[0] - (8 if stress else 0), pc_range[2] + (8 if stress else 0), n)
        bx[:, 1] = rng.uniform(pc_range[1] - (8 if stress else 0), pc_range[3] + (8 if stress else 0), n)
        bx[:, 2] = rng.uniform(-3, 1, n)
        bx[:, 3] = rng.uniform(0.4, 3.0, n)
        bx[:, 4] = rng.uniform(0.4, 12.0, n)
        bx[:, 5] = rng.uniform(0.5, 4.0, n)
        bx[:, 6] = rng.uniform(-np.pi, np.pi, n)
        bx[:, 7:9] = rng.normal(0, 2, (n, 2))
        if stress and n >= 4:
            bx[0, 0] = pc_range[0] - 0.2          # cell coordinate in (-1, 0): truncates to cell 0
            bx[1, 1] = pc_range[1] - 0.3
            bx[2, 3] = 0.0                        # non-positive width
            bx[3, 0:2] = bx[n - 1, 0:2]           # two boxes on one cell
        lb = rng.integers(0, num_labels, n).astype(np.int64)
        if stress and n >= 4:
            lb[3] = lb[n - 1]                     # ... of the same class, so they share a cell of one task
        boxes.append(torch.from_numpy(bx))
        labels.append(torch.from_numpy(lb))
    return boxes, labels


def checkpoint_like_(model, seed=0):
    """Weights shaped like a TRAINED checkpoint rather than a fresh initialisation, on top of
    ``randomize_norm_stats_``:

    * the last BatchNorm of every residual block: gamma drawn from U(0.05, 0.5) per channel -- mmdet zero-initialises it
      (``zero_init_residual``) and training grows it to a fraction of 1;
    * every (convolution, BatchNorm) pair: per-channel raw scales spread over FOUR decades -- ``running_var`` log-uniform in
      [1e-2, 1e2] times its value, with the convolution's output-channel weights (and bias) and ``running_mean`` scaled by the
      square root, as the statistics of a trained network follow the scale of the channel that produces them.  Mathematically the
      network function is unchanged (BatchNorm divides the scale out again), so activations stay O(1-10) and an absolute parity bar
      keeps its meaning; numerically the folded scales 1 / sqrt(var) now range over 0.1 .. 10 and the packed weights over the same
      two decades, which default-initialised statistics (var ~ 1) never exercise.

    Pairs are found by module order: a BatchNorm2d directly after a Conv2d / ConvTranspose2d with as many output channels."""
    from .layers import blocks as _blocks
    randomize_norm_stats_(model, seed)
    g = torch.Generator().manual_seed(1000 + seed)
    last_bn = set()
    for m in model.modules():
        if isinstance(m, _blocks.Bottleneck):
            last_bn.add(m.bn3)
        elif isinstance(m, _blocks.BasicBlock):
            last_bn.add(m.bn2)
    prev = None
    pairs = 0
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                prev = m
            elif isinstance(m, torch.nn.BatchNorm2d):
                if m in last_bn:
                    m.weight.copy_(0.05 + 0.45 * torch.rand(m.num_features, generator=g))
                    m.weight.mul_(torch.where(torch.rand(m.num_features, generator=g) < 0.5, -1.0, 1.0))
                conv = prev
                prev = None
                if conv is None or conv.groups != 1:
                    continue
                transposed = isinstance(conv, torch.nn.ConvTranspose2d)
                cout = conv.weight.shape[1] if transposed else conv.weight.shape[0]
                if cout != m.num_features:
                    continue
                v = 10.0 ** (4.0 * torch.rand(cout, generator=g) - 2.0)          # log-uniform over [1e-2, 1e2]
                r = v.sqrt()
                conv.weight.mul_(r.view(1, -1, 1, 1) if transposed else r.view(-1, 1, 1, 1))
                if conv.bias is not None:
                    conv.bias.mul_(r)
                m.running_mean.mul_(r)
                m.running_var.mul_(v)
                pairs += 1
            elif not isinstance(m, (torch.nn.ReLU, torch.nn.Identity)) and len(list(m.children())) == 0:
                prev = None          # something else between a convolution and a norm: not a pair
    assert pairs > 20, pairs
    return model

