"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by sgv3d_amd/).

Plain PyTorch-CPU fp32 restatement of the reference's camera->BEV forward, written as pure
functions over a ``state_dict`` (reference parameter names) and the reference's
config dicts.  This is BASELINE config 1 ("CPU forward via models/bev_height.py, plumbing, no GPU")
and the fp32 reference the HIP model is compared with.

Follows, line by line where the source is in the reference:
  models/bev_height.py:42-80                      bevheight_forward
  layers/backbones/lss_fpn.py:403-414            get_cam_feats  -> resnet + secondfpn
  layers/backbones/lss_fpn.py:207-250            heightnet (+ ASPP :96-113, SELayer :155-159, Mlp :138-144)
  layers/backbones/lss_fpn.py:462-495            lift, geometry, quantise, voxel pooling
  layers/heads/bev_height_head.py:85-111         head trunk / neck / CenterHead forward
and, for the third-party layers whose source is NOT in the reference (mmdet 2.19.0 ResNet /
BasicBlock / Bottleneck, mmdet3d 0.18.1 SECONDFPN / CenterHead / SeparateHead, mmcv-full 1.4.0
DeformConv2dPack), the published definitions summarised in SURVEY.md §2.2.  PARITY UNPINNED for
those: the reference has no tests or fixtures at that boundary and the packages are absent from
this image; they are checked structurally (tests/test_oracle_model_cpu.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import geometry_ref as G
from . import voxel_pooling_ref as VP

RESNET_BLOCKS = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
                 101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}


BN_TRAINING = False     # True: batch statistics (the training-mode checker, bevheight_train_forward); running stats untouched
BN_FROZEN = ()          # name prefixes of BatchNorm layers that keep their running statistics in training mode (frozen_stages)


def frozen_prefixes(prefix, cfg):
    """mmdet 2.19.0 ``ResNet._freeze_stages`` as name prefixes: with ``frozen_stages >= 0`` the stem (conv1, bn1) is constant and
    bn1 stays in eval mode; stages 1..frozen_stages likewise (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:48:
    ``frozen_stages=0``).  Parameters under these prefixes receive no gradient in the reference's training step."""
    k = cfg.get('frozen_stages', -1)
    if k < 0:
        return ()
    return tuple([f'{prefix}.conv1.', f'{prefix}.bn1.'] + [f'{prefix}.layer{i}.' for i in range(1, k + 1)])


def bn(sd, p, x, eps=1e-5):
    training = BN_TRAINING and not (p + '.').startswith(BN_FROZEN) if BN_FROZEN else BN_TRAINING
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training, 0.0, eps)


def conv(sd, p, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding, dilation)


def basic_block(sd, p, x, stride):
    out = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x, stride, 1)))
    out = bn(sd, p + '.bn2', conv(sd, p + '.conv2', out, 1, 1))
    identity = x
    if p + '.downsample.0.weight' in sd:
        identity = bn(sd, p + '.downsample.1', conv(sd, p + '.downsample.0', x, stride))
    return F.relu(out + identity)


def bottleneck(sd, p, x, stride):
    out = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x)))
    out = F.relu(bn(sd, p + '.bn2', conv(sd, p + '.conv2', out, stride, 1)))        # style='pytorch'
    out = bn(sd, p + '.bn3', conv(sd, p + '.conv3', out))
    identity = x
    if p + '.downsample.0.weight' in sd:
        identity = bn(sd, p + '.downsample.1', conv(sd, p + '.downsample.0', x, stride))
    return F.relu(out + identity)


def resnet(sd, p, x, cfg, use_maxpool=True):
    """mmdet ResNet.forward (stem conv7x7 s2 + BN + ReLU [+ maxpool], stages, out_indices)."""
    kind, blocks = RESNET_BLOCKS[cfg['depth']]
    num_stages = cfg.get('num_stages', 4)
    strides = cfg.get('strides', (1, 2, 2, 2))
    out_indices = cfg.get('out_indices', (0, 1, 2, 3))
    x = F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x, 2, 3)))
    if use_maxpool:
        x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    blk = basic_block if kind == 'basic' else bottleneck
    for i in range(num_stages):
        for j in range(blocks[i]):
            x = blk(sd, f'{p}.layer{i + 1}.{j}', x, strides[i] if j == 0 else 1)
        if i in out_indices:
            outs.append(x)
    return outs


def secondfpn(sd, p, feats, cfg):
    ups = []
    for i, s in enumerate(cfg['upsample_strides']):
        w = sd[f'{p}.deblocks.{i}.0.weight']
        if s >= 1:
            y = F.conv_transpose2d(feats[i], w, None, stride=int(s))
        else:
            k = int(round(1 / s))
            y = F.conv2d(feats[i], w, None, stride=k)
        ups.append(F.relu(bn(sd, f'{p}.deblocks.{i}.1', y, eps=1e-3)))
    return torch.cat(ups, 1)


def deform_conv3x3(x, offset, weight, groups):
    """mmcv 1.4.0 DeformConv2dPack semantics (DCNv1, deform_groups=1, stride 1, pad 1, dil 1):
    offset channel 2t = dy, 2t+1 = dx of tap t (row-major); bilinear sampling with zero padding,
    samples outside (-1, H) x (-1, W) are zero; grouped 3x3 weights, no bias."""
    B, C, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype, device=x.device), torch.arange(W, dtype=x.dtype, device=x.device),
                            indexing="ij")
    cols = []
    flat = x.reshape(B, C, -1)
    for t in range(9):
        ky, kx = t // 3, t % 3
        hf = ys[None] - 1 + ky + offset[:, 2 * t]
        wf = xs[None] - 1 + kx + offset[:, 2 * t + 1]
        valid = (hf > -1) & (wf > -1) & (hf < H) & (wf < W)
        h0, w0 = torch.floor(hf), torch.floor(wf)
        lh, lw = hf - h0, wf - w0
        val = torch.zeros_like(x)
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = (h0 + dh).long(), (w0 + dw).long()
            ok = valid & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1))[:, None].expand(B, C, H, W).reshape(B, C, -1)
            val = val + torch.gather(flat, 2, idx).reshape(B, C, H, W) * (wt * ok)[:, None]
        cols.append(val)
    col = torch.stack(cols, 2)                                           # [B, C, 9, H, W]
    cout = weight.shape[0]
    cpg, opg = C // groups, cout // groups
    out = x.new_zeros(B, cout, H, W)
    for g in range(groups):
        wg = weight[g * opg:(g + 1) * opg].reshape(opg, cpg * 9)
        cg = col[:, g * cpg:(g + 1) * cpg].reshape(B, cpg * 9, H * W)
        out[:, g * opg:(g + 1) * opg] = (wg @ cg).reshape(B, opg, H, W)
    return out


def mlp_input(mats):
    """lss_fpn.py:208-240"""
    intrins = mats['intrin_mats'][:, 0:1, ..., :3, :3]
    B = intrins.shape[0]
    N = intrins.shape[2]
    ida = mats['ida_mats'][:, 0:1, ...]
    s2e = mats['sensor2ego_mats'][:, 0:1, ..., :3, :]
    bda = mats['bda_mat'].view(B, 1, 1, 4, 4).repeat(1, 1, N, 1, 1)
    v = torch.cat([torch.stack([
        intrins[:, 0:1, ..., 0, 0], intrins[:, 0:1, ..., 1, 1], intrins[:, 0:1, ..., 0, 2], intrins[:, 0:1, ..., 1, 2],
        ida[:, 0:1, ..., 0, 0], ida[:, 0:1, ..., 0, 1], ida[:, 0:1, ..., 0, 3],
        ida[:, 0:1, ..., 1, 0], ida[:, 0:1, ..., 1, 1], ida[:, 0:1, ..., 1, 3],
        bda[:, 0:1, ..., 0, 0], bda[:, 0:1, ..., 0, 1], bda[:, 0:1, ..., 1, 0], bda[:, 0:1, ..., 1, 1],
        bda[:, 0:1, ..., 2, 2]], dim=-1), s2e.reshape(B, 1, N, -1)], -1)
    return v.reshape(-1, v.shape[-1])


def _mlp(sd, p, x):
    return F.linear(F.relu(F.linear(x, sd[p + '.fc1.weight'], sd[p + '.fc1.bias'])), sd[p + '.fc2.weight'], sd[p + '.fc2.bias'])


def _se(sd, p, x, x_se):
    g = conv(sd, p + '.conv_expand', F.relu(conv(sd, p + '.conv_reduce', x_se)))
    return x * torch.sigmoid(g)


def aspp(sd, p, x):
    """lss_fpn.py:96-113 (eval: dropout is the identity)."""
    x1 = F.relu(bn(sd, p + '.aspp1.bn', conv(sd, p + '.aspp1.atrous_conv', x)))
    outs = [x1]
    for i, d in ((2, 6), (3, 12), (4, 18)):
        outs.append(F.relu(bn(sd, f'{p}.aspp{i}.bn', conv(sd, f'{p}.aspp{i}.atrous_conv', x, 1, d, d))))
    x5 = F.adaptive_avg_pool2d(x, (1, 1))
    x5 = F.relu(bn(sd, p + '.global_avg_pool.2', conv(sd, p + '.global_avg_pool.1', x5)))
    x5 = F.interpolate(x5, size=x.shape[2:], mode='bilinear', align_corners=True)
    x = torch.cat(outs + [x5], 1)
    return F.relu(bn(sd, p + '.bn1', conv(sd, p + '.conv1', x)))


def heightnet(sd, p, x, mats):
    """lss_fpn.py:207-250 -> [B*N, D + C, fH, fW]"""
    v = mlp_input(mats).to(x.dtype)
    v = F.batch_norm(v, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'],
                     BN_TRAINING, 0.0, 1e-5)
    x = F.relu(bn(sd, p + '.reduce_conv.1', conv(sd, p + '.reduce_conv.0', x, 1, 1)))
    context = _se(sd, p + '.context_se', x, _mlp(sd, p + '.context_mlp', v)[..., None, None])
    context = conv(sd, p + '.context_conv', context)
    h = _se(sd, p + '.height_se', x, _mlp(sd, p + '.height_mlp', v)[..., None, None])
    for i in range(3):
        h = basic_block(sd, f'{p}.height_conv.{i}', h, 1)
    h = aspp(sd, p + '.height_conv.3', h)
    offset = conv(sd, p + '.height_conv.4.conv_offset', h, 1, 1)
    h = deform_conv3x3(h, offset, sd[p + '.height_conv.4.weight'], groups=4)
    height = conv(sd, p + '.height_layer', h)
    return torch.cat([height, context], 1)


def geometry_indices(sd, mats, sweep=0):
    """get_geometry + quantise for every (batch, camera) via the numpy oracle.
    -> int32 [B, N, D, fH, fW, 3]"""
    fr = sd['backbone.frustum'].float().cpu().numpy()
    vc, vs = sd['backbone.voxel_coord'].float().cpu().numpy(), sd['backbone.voxel_size'].float().cpu().numpy()
    mats = {k: v.float().cpu() for k, v in mats.items()}
    B, N = mats['sensor2ego_mats'].shape[0], mats['sensor2ego_mats'].shape[2]
    out = np.empty((B, N) + fr.shape[:3] + (3,), np.int32)
    for b in range(B):
        for n in range(N):
            gi, _ = G.geom_xyz_for_camera(
                fr, mats['sensor2ego_mats'][b, sweep, n].numpy(), mats['sensor2virtual_mats'][b, sweep, n].numpy(),
                mats['intrin_mats'][b, sweep, n].numpy(), mats['ida_mats'][b, sweep, n].numpy(),
                float(mats['reference_heights'][b, sweep, n]), mats['bda_mat'][b].numpy() if 'bda_mat' in mats else None,
                vc, vs)
            out[b, n] = gi
    return out


def lss_fpn_forward_sweeps(sd, conf, imgs, mats):
    """LSSFPN.forward with num_sweeps > 1 (lss_fpn.py:535-550): every sweep through _forward_single_sweep with its own
    images and its own geometry (mats[:, sweep_index]); HeightNet reads the KEY frame's intrinsics / ida / sensor2ego for
    every sweep (its forward indexes ``[:, 0:1]``, lss_fpn.py:208-212); BEV maps concatenated on the channel axis."""
    return torch.cat([lss_fpn_forward(sd, conf, imgs, mats, sweep=s) for s in range(imgs.shape[1])], 1)


def lss_fpn_forward(sd, conf, imgs, mats, keep=None, cam_feats=None, sweep=0):
    """LSSFPN._forward_single_sweep (lss_fpn.py:422-495), one sweep -> BEV [B, C, Y, X].
    ``cam_feats`` [B*N, C, fH, fW]: stands in for get_cam_feats (:403-414) -- the pinned-composition test feeds the
    same neck features the golden run of the reference was given (tests/golden/make_golden_modules.py)."""
    B, S, N, Cin, H, W = imgs.shape
    if cam_feats is None:
        x = imgs[:, sweep].reshape(B * N, Cin, H, W)
        feats = resnet(sd, 'backbone.img_backbone', x, conf['img_backbone_conf'])
        src = secondfpn(sd, 'backbone.img_neck', feats, conf['img_neck_conf'])
    else:
        feats, src = None, cam_feats
    hf = heightnet(sd, 'backbone.height_net', src, mats)
    D = sd['backbone.frustum'].shape[0]
    C = conf['output_channels']
    height = hf[:, :D].softmax(1)                                                        # :462
    lifted = height.unsqueeze(1) * hf[:, D:D + C].unsqueeze(2)                           # :464-466
    fH, fW = lifted.shape[3], lifted.shape[4]
    lifted = lifted.reshape(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2).contiguous()   # :469-486
    geom = geometry_indices(sd, mats, sweep)                                             # :478-488
    vn = [int(v) for v in sd['backbone.voxel_num']]
    bev, _ = VP.forward(geom, lifted.numpy(), vn, want_pos_memo=False)                   # :490-491
    bev = torch.from_numpy(bev)
    if keep is not None:
        keep.update(img_feats=src, height_feature=hf, geom_xyz=geom, bev=bev, backbone_feats=feats)
    return bev


# ------------------------------------------------------------------------------------------------
# SGV3D BSM variant (layers/backbones/bsm_lss_fpn.py)
# ------------------------------------------------------------------------------------------------
def _sablock(sd, p, x, y):
    """bsm_lss_fpn.py:151-160"""
    return conv(sd, p + '.conv', x, 1, 1) * torch.sigmoid(conv(sd, p + '.attention.0', y, 1, 1))


def _task_decoder(sd, p, x):
    """TaskHead.decoder, bsm_lss_fpn.py:184-190"""
    x = basic_block(sd, p + '.decoder.0', x, 1)
    x = basic_block(sd, p + '.decoder.1', x, 1)
    return F.relu(bn(sd, p + '.decoder.3', conv(sd, p + '.decoder.2', x, 1, 1)))


def _task_fpn(sd, p, feat0, feat1):
    """bsm_lss_fpn.py:209-212"""
    feat0 = conv(sd, p + '.reduce_conv', F.interpolate(feat0, scale_factor=2, mode='bilinear'), 1, 1)
    return feat0 + _sablock(sd, p + '.self_attention', feat1, feat0)


def msct_head(sd, p, feats, mats):
    """MSCThead.forward, bsm_lss_fpn.py:259-320 -> (depth1, semantic1, context1, semantic0)"""
    v = mlp_input(mats)
    v = F.batch_norm(v.to(feats[0].dtype), sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'],
                     sd[p + '.bn.bias'], BN_TRAINING, 0.0, 1e-5)
    s0 = F.relu(bn(sd, p + '.reduce_conv0.1', conv(sd, p + '.reduce_conv0.0', feats[0], 1, 1)))
    s1 = F.relu(bn(sd, p + '.reduce_conv1.1', conv(sd, p + '.reduce_conv1.0', feats[1], 1, 1)))
    s0 = _se(sd, p + '.scale0_se', s0, _mlp(sd, p + '.scale0_mlp', v)[..., None, None])
    s1 = _se(sd, p + '.scale1_se', s1, _mlp(sd, p + '.scale1_mlp', v)[..., None, None])
    s0 = aspp(sd, p + '.aspp', s0)
    depth_feat = s0                                               # TaskHead(with_head=False) returns its input
    semantic_feat = _task_decoder(sd, p + '.semantic_head0', s0)
    semantic0 = conv(sd, p + '.semantic_head0.head', semantic_feat)
    context_feat = F.relu(bn(sd, p + '.context_conv0.1', conv(sd, p + '.context_conv0.0', s0, 1, 1)))
    depth_feat = _task_fpn(sd, p + '.depth_fpn', depth_feat, s1)
    semantic_feat = _task_fpn(sd, p + '.semantic_fpn', semantic_feat, s1)
    context_feat = _task_fpn(sd, p + '.context_fpn', context_feat, s1)
    depth1 = conv(sd, p + '.depth_head1.head', _task_decoder(sd, p + '.depth_head1', depth_feat))
    semantic1 = conv(sd, p + '.semantic_head1.head', _task_decoder(sd, p + '.semantic_head1', semantic_feat))
    c = F.relu(bn(sd, p + '.context_conv1.1', conv(sd, p + '.context_conv1.0', context_feat, 1, 1)))
    context1 = conv(sd, p + '.context_conv1.3', c)
    return depth1, semantic1, context1, semantic0


def bsm_lss_fpn_forward(sd, conf, imgs, mats, keep=None, cam_feats=None):
    """BSMLSSFPN._forward_single_sweep (bsm_lss_fpn.py:485-559) -> BEV [B, 87, Y, X].
    ``cam_feats`` = (stride-16 features, stride-8 features), each [B*N, C, h, w]: stands in for get_cam_feats."""
    B, S, N, Cin, H, W = imgs.shape
    if cam_feats is None:
        x = imgs[:, 0].reshape(B * N, Cin, H, W)
        feats = resnet(sd, 'backbone.img_backbone', x, conf['img_backbone_conf'])
        n16 = secondfpn(sd, 'backbone.img_neck_16', feats, conf['img_neck_conf'])
        cfg8 = dict(conf['img_neck_conf'], upsample_strides=[0.5, 1, 2, 4])            # :368
        n8 = secondfpn(sd, 'backbone.img_neck_8', feats, cfg8)
    else:
        n16, n8 = cam_feats
    depth1, semantic1, context1, semantic0 = msct_head(sd, 'backbone.height_net', [n16, n8], mats)
    height = depth1.softmax(dim=1)                                                    # :521
    semantic = semantic1.softmax(dim=1)
    tran = torch.cat((context1, semantic), dim=1)                                     # :524
    mask = semantic[:, 0, :, :].unsqueeze(1) > 0.45                                   # :526 background
    tran = tran * (1 - mask.int())
    lifted = height.unsqueeze(1) * tran.unsqueeze(2)                                  # :530
    C, D, fH, fW = lifted.shape[1:]
    lifted = lifted.reshape(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2).contiguous()
    geom = geometry_indices(sd, mats)
    vn = [int(v) for v in sd['backbone.voxel_num']]
    bev, _ = VP.forward(geom, lifted.numpy(), vn, want_pos_memo=False)
    bev = torch.from_numpy(bev)
    if keep is not None:
        keep.update(neck16=n16, neck8=n8, depth1=depth1, semantic1=semantic1, context1=context1, geom_xyz=geom, bev=bev)
    return bev


def head_forward(sd, conf, x, keep=None):
    """BEVHeightHead.forward (bev_height_head.py:85-111) + mmdet3d CenterHead.forward."""
    bcfg = conf.get('bev_backbone_conf')
    ncfg = conf.get('bev_neck_conf')
    trunk_outs = [x] + resnet(sd, 'head.trunk', x, bcfg, use_maxpool=False)
    fpn = secondfpn(sd, 'head.neck', trunk_outs, ncfg)
    shared = F.relu(bn(sd, 'head.shared_conv.bn', conv(sd, 'head.shared_conv.conv', fpn, 1, 1)))
    branches = list(conf['common_heads'].keys()) + ['heatmap']
    ret = []
    for t in range(len(conf['tasks'])):
        d = {}
        for name in branches:
            p = f'head.task_heads.{t}.{name}'
            h = F.relu(bn(sd, p + '.0.bn', conv(sd, p + '.0.conv', shared, 1, 1)))
            d[name] = conv(sd, p + '.1', h, 1, 1)
        ret.append([d])
    if keep is not None:
        keep.update(fpn=fpn, shared=shared)
    return tuple(ret)


def voxel_pool_torch(geom, lifted, voxel_num):
    """Differentiable restatement of the scatter (voxel_pooling_forward_cuda.cu:9-36): out-of-range points are dropped,
    the others add their feature row to cell (b, y, x).  geom int32 numpy [B, ..., 3], lifted [B, ..., C] -> [B, C, Y, X]."""
    X, Y, Z = voxel_num
    B, C = lifted.shape[0], lifted.shape[-1]
    g = torch.from_numpy(np.ascontiguousarray(geom)).reshape(B, -1, 3).long().to(lifted.device)
    f = lifted.reshape(B, -1, C)
    ok = (g[..., 0] >= 0) & (g[..., 0] < X) & (g[..., 1] >= 0) & (g[..., 1] < Y) & (g[..., 2] >= 0) & (g[..., 2] < Z)
    cell = (torch.arange(B, device=lifted.device)[:, None] * Y + g[..., 1]) * X + g[..., 0]
    out = f.new_zeros(B * Y * X, C).index_add(0, cell[ok], f[ok])
    return out.reshape(B, Y, X, C).permute(0, 3, 1, 2)


def bevheight_train_forward(sd, backbone_conf, head_conf, imgs, mats, is_train_height=False):
    """BEVHeight.forward in training mode (BatchNorm on batch statistics, models/bev_height.py:42-80) WITH an autograd
    graph over ``sd``: the checker of sgv3d_amd/train_forward.py, for LSSFPN and the BSM variant.  With
    ``is_train_height`` returns ``(preds, height_pred)`` (:72-77): (semantic0, semantic1) logits of BSMLSSFPN
    (bsm_lss_fpn.py:557-558) / (assist, assist) of LSSFPN (lss_fpn.py:459,493-494).  Dropout is not restated (the tests
    set p = 0).  ``frozen_stages`` of the image backbone's config is honoured (``frozen_prefixes``): those BatchNorm layers use
    their running statistics, and the caller leaves ``requires_grad`` off for the parameters under those prefixes."""
    global BN_TRAINING, BN_FROZEN
    BN_TRAINING = True
    BN_FROZEN = frozen_prefixes('backbone.img_backbone', backbone_conf['img_backbone_conf'])
    try:
        B, S, N, Cin, H, W = imgs.shape
        dt = next(iter(sd.values())).dtype
        x = imgs[:, 0].reshape(B * N, Cin, H, W).to(dt)
        feats = resnet(sd, 'backbone.img_backbone', x, backbone_conf['img_backbone_conf'])
        geo_sd = {k: v.detach().float() for k, v in sd.items() if k.startswith('backbone.frustum') or
                  k.startswith('backbone.voxel')}
        if backbone_conf.get('is_bsm'):
            n16 = secondfpn(sd, 'backbone.img_neck_16', feats, backbone_conf['img_neck_conf'])
            n8 = secondfpn(sd, 'backbone.img_neck_8', feats, dict(backbone_conf['img_neck_conf'], upsample_strides=[0.5, 1, 2, 4]))
            depth1, semantic1, context1, semantic0 = msct_head(sd, 'backbone.height_net', [n16, n8], mats)
            semantic = semantic1.softmax(dim=1)
            tran = torch.cat((context1, semantic), dim=1)
            tran = tran * (1 - (semantic[:, 0:1] > 0.45).int())
            lifted = depth1.softmax(dim=1).unsqueeze(1) * tran.unsqueeze(2)
            aux = (semantic0, semantic1)
        else:
            src = secondfpn(sd, 'backbone.img_neck', feats, backbone_conf['img_neck_conf'])
            hf = heightnet(sd, 'backbone.height_net', src, mats)
            D, C = sd['backbone.frustum'].shape[0], backbone_conf['output_channels']
            lifted = hf[:, :D].softmax(1).unsqueeze(1) * hf[:, D:D + C].unsqueeze(2)
            assist = conv(sd, 'backbone.assist_layer', src)
            aux = (assist, assist)
        C, D, fH, fW = lifted.shape[1:]
        lifted = lifted.reshape(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
        geom = geometry_indices(geo_sd, mats)
        bev = voxel_pool_torch(geom, lifted, [int(v) for v in sd['backbone.voxel_num']])
        preds = head_forward(sd, head_conf, bev)
        return (preds, aux) if is_train_height else preds
    finally:
        BN_TRAINING = False
        BN_FROZEN = ()


def bevheight_forward(sd, backbone_conf, head_conf, imgs, mats, keep=None):
    """BEVHeight.forward, eval branch (models/bev_height.py:78-80)."""
    with torch.no_grad():
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        mats = {k: v.detach().cpu() for k, v in mats.items()}
        fwd = bsm_lss_fpn_forward if backbone_conf.get('is_bsm') else lss_fpn_forward
        bev = fwd(sd, backbone_conf, imgs.detach().cpu().float(), mats, keep)
        return head_forward(sd, head_conf, bev, keep)


def bevheight_forward_highprec(sd, backbone_conf, head_conf, imgs, mats, device="cpu", dtype=torch.float64, keep=None):
    """The same eval-mode forward evaluated in ``dtype`` (float64) on ``device``: the yardstick that tells how far BOTH
    float32 executions -- this oracle on torch-CPU and the HIP path -- are from exact arithmetic (two float32
    implementations of a 100-layer network differ from each other by their summed rounding noise, so the meaningful
    question is whether the HIP path is as close to the exact result as the reference-style float32 execution is).
    Voxel indices come from the float32 geometry oracle (they are integer data, identical by construction); the splat is
    an index_add in ``dtype``.  A checker like the rest of this file; on a GPU it runs through torch's im2col + dgemm
    convolution path."""
    with torch.no_grad():
        sdd = {k: (v.detach().to(device=device, dtype=dtype) if v.is_floating_point() else v.detach().to(device))
               for k, v in sd.items()}
        matsd = {k: v.detach().to(device=device, dtype=dtype) for k, v in mats.items()}
        B, S, N, Cin, H, W = imgs.shape
        x = imgs[:, 0].reshape(B * N, Cin, H, W).to(device=device, dtype=dtype)
        feats = resnet(sdd, 'backbone.img_backbone', x, backbone_conf['img_backbone_conf'])
        if backbone_conf.get('is_bsm'):
            n16 = secondfpn(sdd, 'backbone.img_neck_16', feats, backbone_conf['img_neck_conf'])
            n8 = secondfpn(sdd, 'backbone.img_neck_8', feats, dict(backbone_conf['img_neck_conf'], upsample_strides=[0.5, 1, 2, 4]))
            depth1, semantic1, context1, _ = msct_head(sdd, 'backbone.height_net', [n16, n8], matsd)
            semantic = semantic1.softmax(dim=1)
            tran = torch.cat((context1, semantic), dim=1)
            tran = tran * (1 - (semantic[:, 0:1] > 0.45).to(dtype))
            lifted = depth1.softmax(dim=1).unsqueeze(1) * tran.unsqueeze(2)
            src = hf = None
        else:
            src = secondfpn(sdd, 'backbone.img_neck', feats, backbone_conf['img_neck_conf'])
            hf = heightnet(sdd, 'backbone.height_net', src, matsd)
            D, C = sdd['backbone.frustum'].shape[0], backbone_conf['output_channels']
            lifted = hf[:, :D].softmax(1).unsqueeze(1) * hf[:, D:D + C].unsqueeze(2)
        C, D, fH, fW = lifted.shape[1:]
        lifted = lifted.reshape(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
        geom = geometry_indices(sd, mats)
        bev = voxel_pool_torch(geom, lifted, [int(v) for v in sd['backbone.voxel_num']])
        if keep is not None:
            keep.update(img_feats=src, height_feature=hf, bev=bev, geom_xyz=geom)
        return head_forward(sdd, head_conf, bev)
