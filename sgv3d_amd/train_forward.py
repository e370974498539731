"""Training-mode forward of ``BEVHeight``, differentiable end to end.

The reference trains the same modules through autograd with cuDNN behind every convolution
(exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:224-240).  Here the layers that carry the flops and
the HBM traffic run on the hand-written kernels, forward and backward:

* every nn.Conv2d / nn.ConvTranspose2d -> ``conv_grad.conv2d`` / ``conv_transpose2d`` (MFMA implicit GEMM / Winograd
  forward, data gradient through the same kernels, MFMA weight-gradient kernel);
* the frustum geometry -> ``sgv3d_geometry_voxel_index`` (integer indices, no gradient);
* the splat -> the ``voxel_pooling`` autograd operator (planned gather forward, gather backward);
* targets / loss / optimiser -> ``BEVHeightHead.get_targets`` / ``.loss`` / ``train_step.DataParallelAdamW``.

* BatchNorm with batch statistics + residual add + ReLU -> ``norm_grad.batch_norm_act`` (one statistics pass and one
  apply pass forward, one reduction pass and one apply pass backward; running statistics updated in the kernel).

* the BSM variant's x2 bilinear upsampling and spatial-attention gate -> ``bsm_grad`` (forward kernels of the
  inference path, adjoint kernels in csrc/bsm_train.hip).

* max pooling of the stem, the ASPP image-pooling branch (global average + 1x1 convolution) and the deformable bilinear
  sampling of the DCN -> ``misc_grad`` (forward kernels of the inference path or their index-keeping variants, adjoint
  kernels in csrc/train_misc.hip; round 2).

The small layers in between -- the 27-feature MLPs with their BatchNorm1d, the softmax over height bins / semantic
classes, concatenations, elementwise gates -- use torch operators on the same NHWC buffers.  Activations are NHWC float32
throughout (an NCHW view with channels-last strides is handed to the torch operators, no layout copies).

Layer semantics follow the eval-mode HIP path module by module (layers/blocks.py, layers/backbones/lss_fpn.py,
layers/heads/bev_height_head.py); the only differences are the ones training mode implies in the reference:
BatchNorm uses and updates batch statistics -- except in the stages ``frozen_stages`` freezes (the image backbone's stem in every
shipped config), which keep their running statistics and receive no gradient --, Dropout(0.5) in the ASPP is active.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import conv_grad, hip_ops
from .bsm_grad import add_mul_sigmoid, upsample_bilinear2x
from . import misc_grad
from .norm_grad import batch_norm_act, deferred_counters
from .layers import blocks
from .layers.backbones import bsm_lss_fpn, lss_fpn
from .ops.voxel_pooling import voxel_pooling

__all__ = ['bevheight_train_forward']

HEAD_CONCAT = os.environ.get("SGV3D_HEAD_CONCAT", "1") != "0"        # 0: the CenterHead branches as n separate maps (diagnostic)
THIN_BATCHED = os.environ.get("SGV3D_THIN_BATCHED", "1") != "0"      # 0: the CenterHead's final layers one by one (diagnostic)


def _nchw(x):            # NHWC tensor -> NCHW view (channels-last strides)
    return x.permute(0, 3, 1, 2)


def _nhwc(x):            # NCHW tensor (any strides) -> contiguous NHWC
    return x.permute(0, 2, 3, 1).contiguous()


def conv(m, x):
    return conv_grad.conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0], m.dilation[0])


def bn(m, x, relu=False, residual=None):
    """``relu(norm(x) + residual)``: one fused pass (csrc/bn_train.hip) for a BatchNorm2d in training mode; a module
    kept in eval mode (norm_eval) or with an odd channel count goes through torch."""
    if m.training and m.track_running_stats and x.shape[-1] % 4 == 0:
        return batch_norm_act(m, x.contiguous(), None if residual is None else residual.contiguous(), relu)
    y = _nhwc(F.batch_norm(_nchw(x), m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps))
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def basic_block(b, x):
    identity = x if b.downsample is None else bn(b.downsample[1], conv(b.downsample[0], x))
    out = bn(b.bn1, conv(b.conv1, x), relu=True)
    return bn(b.bn2, conv(b.conv2, out), relu=True, residual=identity)


def bottleneck(b, x):
    identity = x if b.downsample is None else bn(b.downsample[1], conv(b.downsample[0], x))
    out = bn(b.bn1, conv(b.conv1, x), relu=True)
    out = bn(b.bn2, conv(b.conv2, out), relu=True)
    return bn(b.bn3, conv(b.conv3, out), relu=True, residual=identity)


def block(b, x):
    return bottleneck(b, x) if isinstance(b, blocks.Bottleneck) else basic_block(b, x)


def _frozen_stem(r, x):
    """conv1 + bn1 + ReLU of a ResNet whose stem is frozen (mmdet ``frozen_stages >= 0``: the reference's image backbone,
    exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:48): constants of the step, so the inference form runs -- one
    convolution with the running statistics folded into its epilogue, no statistics pass, no graph node, no weight gradient.  The
    packed stem is rebuilt when one of its five tensors was written or replaced (``load_state_dict``)."""
    ts = (r.conv1.weight, r.bn1.weight, r.bn1.bias, r.bn1.running_mean, r.bn1.running_var)
    tag = tuple((t.data_ptr(), t._version) for t in ts) + (hip_ops.switch_state(),)
    if getattr(r, '_frozen_stem_tag', None) != tag:
        r._hip, r._frozen_stem_tag = None, tag
    with torch.no_grad():
        return r.hip_stem(x)


def resnet(r, x, use_maxpool=True):
    x = _frozen_stem(r, x) if r.frozen_stem() else bn(r.bn1, conv(r.conv1, x), relu=True)
    if use_maxpool:
        x = misc_grad.maxpool3x3s2(x)
    outs = []
    for i, name in enumerate(r.res_layers):
        for b in getattr(r, name):
            x = block(b, x)
        if i in r.out_indices:
            outs.append(x)
    return outs


def secondfpn(n, feats):
    ups = []
    for blk, f in zip(n.deblocks, feats):
        layer = blk[0]
        if isinstance(layer, nn.ConvTranspose2d):
            y = conv_grad.conv_transpose2d(f, layer.weight, layer.stride[0])
        else:
            y = conv(layer, f)
        ups.append(bn(blk[1], y, relu=True))
    return torch.cat(ups, -1)


def aspp(a, x):
    branches = [bn(m.bn, conv(m.atrous_conv, x), relu=True) for m in (a.aspp1, a.aspp2, a.aspp3, a.aspp4)]
    g = a.global_avg_pool                                                  # AdaptiveAvgPool2d((1, 1)) + 1x1 conv: HIP kernels
    x5 = bn(g[2], misc_grad.pooled_linear(x, g[1].weight)[:, None, None, :], relu=True)
    branches.append(x5.expand(-1, x.shape[1], x.shape[2], -1))             # bilinear upsampling of a 1x1 map = broadcast
    y = bn(a.bn1, conv(a.conv1, torch.cat(branches, -1).contiguous()), relu=True)
    return F.dropout(y, a.dropout.p, a.dropout.training)


def deform_conv(d, x):
    """mmcv DeformConv2dPack (3x3, pad 1, deform_groups 1): deformable bilinear im2col (HIP kernel, differentiable in the
    input and the offsets through its adjoint kernel), then one 1x1 HIP convolution per group over the (tap, channel)
    columns."""
    B, H, W, C = (int(v) for v in x.shape)
    offset = conv(d.conv_offset, x)                                        # [B, H, W, 18]: (dy, dx) per tap
    g = d.groups
    cpg, opg = d.in_channels // g, d.out_channels // g
    col = misc_grad.deform_im2col3x3(x, offset, g)                         # [B, H, W, g * 9 * cpg], HIP forward + adjoint
    outs = []
    for gi in range(g):
        wg = d.weight[gi * opg:(gi + 1) * opg].permute(0, 2, 3, 1).reshape(opg, 9 * cpg, 1, 1)
        cg = col[..., gi * 9 * cpg:(gi + 1) * 9 * cpg].contiguous()
        outs.append(conv_grad.conv2d(cg, wg))
    return torch.cat(outs, -1)


def _gate(mlp, se, v):
    h = mlp.drop2(mlp.fc2(mlp.drop1(mlp.act(mlp.fc1(v)))))
    w_r = se.conv_reduce.weight.reshape(se.conv_reduce.out_channels, -1)
    w_e = se.conv_expand.weight.reshape(se.conv_expand.out_channels, -1)
    h = F.relu(F.linear(h, w_r, se.conv_reduce.bias))
    return torch.sigmoid(F.linear(h, w_e, se.conv_expand.bias))            # [B*N, mid]


def heightnet(hn, x, mats_dict):
    """lss_fpn.py:207-250 -> (height logits [BN, H, W, D], context [BN, H, W, C])."""
    v = hn.bn(lss_fpn.HeightNet.mlp_input(mats_dict))
    x = bn(hn.reduce_conv[1], conv(hn.reduce_conv[0], x), relu=True)
    context = conv(hn.context_conv, x * _gate(hn.context_mlp, hn.context_se, v)[:, None, None, :])
    h = x * _gate(hn.height_mlp, hn.height_se, v)[:, None, None, :]
    for m in hn.height_conv:
        if isinstance(m, lss_fpn.ASPP):
            h = aspp(m, h)
        elif isinstance(m, lss_fpn.DCN):
            h = deform_conv(m, h)
        else:
            h = basic_block(m, h)
    return conv(hn.height_layer, h), context


def lss_fpn_forward(bb, imgs, mats_dict, want_feats=False):
    """LSSFPN._forward_single_sweep for the key frame (lss_fpn.py:422-495) -> BEV map NHWC [B, Y, X, C]
    (and, with ``want_feats``, the neck features the assist layer reads, :459)."""
    B, S, N, _, imH, imW = imgs.shape
    assert S == 1, "one sweep (every shipped config)"
    x = hip_ops.nchw_to_nhwc(imgs.reshape(B * N, 3, imH, imW).float().contiguous(), c_pad=4)
    feats = secondfpn(bb.img_neck, resnet(bb.img_backbone, x))
    height, context = heightnet(bb.height_net, feats, mats_dict)
    prob = height.softmax(-1)                                              # over the D height bins (:483)
    lifted = prob.permute(0, 3, 1, 2).unsqueeze(-1) * context.unsqueeze(1)  # [BN, D, fH, fW, C] (:484-486)
    with torch.no_grad():
        geom = bb.get_geometry_voxel_index(
            mats_dict['sensor2ego_mats'][:, 0], mats_dict['sensor2virtual_mats'][:, 0], mats_dict['intrin_mats'][:, 0],
            mats_dict['ida_mats'][:, 0], mats_dict['reference_heights'][:, 0], mats_dict.get('bda_mat', None))
    D, fH, fW, C = (int(v) for v in lifted.shape[1:])
    bev = voxel_pooling(geom, lifted.reshape(B, N, D, fH, fW, C).contiguous(), bb._voxel_num_host)   # [B, C, Y, X] view
    return (bev.permute(0, 2, 3, 1), feats) if want_feats else bev.permute(0, 2, 3, 1)


# ---------------------------------------------------------------------------------------------------------------
# SGV3D BSM variant (layers/backbones/bsm_lss_fpn.py), SURVEY.md §8(f) rank 3
# ---------------------------------------------------------------------------------------------------------------
def task_decoder(th, x):
    """TaskHead.decoder (bsm_lss_fpn.py:184-190)."""
    x = basic_block(th.decoder[0], x)
    x = basic_block(th.decoder[1], x)
    return bn(th.decoder[3], conv(th.decoder[2], x), relu=True)


def task_fpn(f, feat0, feat1):
    """TaskFPN.forward (bsm_lss_fpn.py:209-212): reduce(upsample(feat0)) + conv(feat1) * sigmoid(attention(that))."""
    feat0 = conv(f.reduce_conv, upsample_bilinear2x(feat0))
    sa = f.self_attention
    return add_mul_sigmoid(feat0, conv(sa.conv, feat1), conv(sa.attention[0], feat0))


def msct_head(hn, feats, mats_dict):
    """MSCThead.forward (bsm_lss_fpn.py:259-320) -> (depth1, semantic1, context1, semantic0), NHWC."""
    v = hn.bn(lss_fpn.HeightNet.mlp_input(mats_dict))
    s0 = bn(hn.reduce_conv0[1], conv(hn.reduce_conv0[0], feats[0]), relu=True)
    s1 = bn(hn.reduce_conv1[1], conv(hn.reduce_conv1[0], feats[1]), relu=True)
    s0 = s0 * _gate(hn.scale0_mlp, hn.scale0_se, v)[:, None, None, :]
    s1 = s1 * _gate(hn.scale1_mlp, hn.scale1_se, v)[:, None, None, :]
    s0 = aspp(hn.aspp, s0)
    depth_feat = s0                                                         # TaskHead(with_head=False) is the identity (:195-199)
    semantic_feat = task_decoder(hn.semantic_head0, s0)
    semantic0 = conv(hn.semantic_head0.head, semantic_feat)
    context_feat = bn(hn.context_conv0[1], conv(hn.context_conv0[0], s0), relu=True)
    depth_feat = task_fpn(hn.depth_fpn, depth_feat, s1)
    semantic_feat = task_fpn(hn.semantic_fpn, semantic_feat, s1)
    context_feat = task_fpn(hn.context_fpn, context_feat, s1)
    depth1 = conv(hn.depth_head1.head, task_decoder(hn.depth_head1, depth_feat))
    semantic1 = conv(hn.semantic_head1.head, task_decoder(hn.semantic_head1, semantic_feat))
    c = bn(hn.context_conv1[1], conv(hn.context_conv1[0], context_feat), relu=True)
    return depth1, semantic1, conv(hn.context_conv1[3], c), semantic0


def bsm_lss_fpn_forward(bb, imgs, mats_dict):
    """BSMLSSFPN._forward_single_sweep for the key frame (bsm_lss_fpn.py:485-559) -> (BEV map NHWC [B, Y, X, 88],
    (semantic0, semantic1) logits as NCHW views -- what the reference returns under is_train_height, :557-558)."""
    B, S, N, _, imH, imW = imgs.shape
    assert S == 1, "one sweep (every shipped config)"
    x = hip_ops.nchw_to_nhwc(imgs.reshape(B * N, 3, imH, imW).float().contiguous(), c_pad=4)
    feats = resnet(bb.img_backbone, x)
    n16, n8 = secondfpn(bb.img_neck_16, feats), secondfpn(bb.img_neck_8, feats)
    depth1, semantic1, context1, semantic0 = msct_head(bb.height_net, [n16, n8], mats_dict)
    height = depth1.softmax(-1)                                            # :521
    semantic = semantic1.softmax(-1)                                       # :522
    tran = torch.cat((context1, semantic), -1)                             # :524
    keep = 1 - (semantic[..., :1] > bb.background_threshold).int()         # :526-527
    tran = tran * keep
    C = int(tran.shape[-1])
    if C % 4:
        tran = F.pad(tran, (0, 4 - C % 4))                                 # 87 -> 88, as the inference path carries it
    lifted = height.permute(0, 3, 1, 2).unsqueeze(-1) * tran.unsqueeze(1)  # [BN, D, fH, fW, 88] (:530)
    with torch.no_grad():
        geom = bb.get_geometry_voxel_index(
            mats_dict['sensor2ego_mats'][:, 0], mats_dict['sensor2virtual_mats'][:, 0], mats_dict['intrin_mats'][:, 0],
            mats_dict['ida_mats'][:, 0], mats_dict['reference_heights'][:, 0], mats_dict.get('bda_mat', None))
    D, fH, fW, Cp = (int(v) for v in lifted.shape[1:])
    bev = voxel_pooling(geom, lifted.reshape(B, N, D, fH, fW, Cp).contiguous(), bb._voxel_num_host)
    return bev.permute(0, 2, 3, 1), (_nchw(semantic0), _nchw(semantic1))


def head_forward(head, bev):
    """BEVHeightHead.forward (bev_height_head.py:85-111) + CenterHead.forward_single."""
    t = head.trunk
    if bev.shape[-1] % 4:
        bev = F.pad(bev, (0, 4 - bev.shape[-1] % 4))
    outs = [bev]
    h = bn(t.bn1, conv(t.conv1, bev.contiguous()), relu=True)
    for i, name in enumerate(t.res_layers):
        for b in getattr(t, name):
            h = block(b, h)
        if i in t.out_indices:
            outs.append(h)
    fpn = secondfpn(head.neck, outs)
    sc = head.shared_conv
    shared = bn(sc.bn, conv(sc.conv, fpn), relu=True)
    # the first layers of all branches read the same map: when they are plain 3x3 / stride 1 / pad 1 convolutions of one shape, their
    # weight gradients run as ONE batched launch (conv_grad.multi_conv2d; 36 launches of one 64 x 64 tile each otherwise)
    seqs = [(th, name, getattr(th, name)) for th in head.task_heads for name in th.heads]
    firsts = [seq[0].conv for _, _, seq in seqs if len(seq) == 2]
    batched = (len(firsts) == len(seqs) and 1 < len(firsts) <= 48 and shared.shape[-1] % 4 == 0 and
               all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.bias is None and
                   c.weight.shape == firsts[0].weight.shape and c.weight.shape[0] % 4 == 0 for c in firsts))
    finals = [seq[-1] for _, _, seq in seqs]
    bns = [seq[0].bn for _, _, seq in seqs] if batched else []
    if (batched and HEAD_CONCAT and all(isinstance(b, nn.BatchNorm2d) and b.training and b.track_running_stats and b.affine and b.momentum is not None
                                        and b.eps == bns[0].eps and b.momentum == bns[0].momentum for b in bns)
            and conv_grad.thin_conv_eligible(finals, [shared.new_empty((1, 1, 1, int(firsts[0].weight.shape[0])))] * len(finals))
            and int(firsts[0].weight.shape[0]) <= 64
            and shared.shape[0] * shared.shape[1] * shared.shape[2] * len(firsts) * int(firsts[0].weight.shape[0]) * 4 < (3 << 30)):      # (the thin kernels address < 3 GiB)
        # the n branches as ONE wide map: a single convolution with the n first-layer weights concatenated (one forward, one data-gradient
        # and one weight-gradient launch instead of n each, and no running sum through n residual epilogues), BatchNorm over the n * 64
        # channels in one launch sequence (it is per channel), the final layers on the map's channel slices (conv_grad.sliced_thin_conv2d)
        from .norm_grad import batch_norm_act_multi
        big = conv_grad.conv2d(shared, conv_grad.cat_params([c.weight for c in firsts]), None, 1, 1, 1)
        hid = batch_norm_act_multi(bns, big, relu=True)
        outs = conv_grad.sliced_thin_conv2d(hid, finals)
        ret, k = [], 0
        for th in head.task_heads:
            d = {}
            for name in th.heads:
                d[name] = _nchw(outs[k])
                k += 1
            ret.append([d])
        return tuple(ret)
    first_out = conv_grad.multi_conv2d(shared, [c.weight for c in firsts]) if batched else None
    hidden, k = [], 0
    for _, _, seq in seqs:
        if batched:
            y = bn(seq[0].bn, first_out[k], relu=True)
            k += 1
        else:
            y = shared
            for layer in seq[:-1]:
                y = bn(layer.bn, conv(layer.conv, y), relu=True)
        hidden.append(y)
    # ... and the final layers (64 -> 1..3 channels) have their whole backward -- data, weight and bias gradients of all branches -- as
    # one batched call (conv_grad.multi_thin_conv2d; three launches per layer otherwise, each a few microseconds of work)
    if THIN_BATCHED and conv_grad.thin_conv_eligible(finals, hidden):
        outs = conv_grad.multi_thin_conv2d(hidden, finals)
    else:
        outs = [conv(f, y) for f, y in zip(finals, hidden)]
    ret, k = [], 0
    for th in head.task_heads:
        d = {}
        for name in th.heads:
            d[name] = _nchw(outs[k])                                       # [B, c, H, W] like the reference
            k += 1
        ret.append([d])
    return tuple(ret)


def bevheight_train_forward(model, imgs, mats_dict):
    """``BEVHeight.forward`` in training mode (models/bev_height.py:42-80): images -> per-task prediction maps with an
    autograd graph; with ``is_train_height`` the tuple ``(preds, height_pred)`` of :72-77, where ``height_pred`` is what
    the backbone hands out beside the BEV map -- ``(semantic0, semantic1)`` logits for the BSM variant
    (bsm_lss_fpn.py:557-558), ``(assist_features, assist_features)`` for ``LSSFPN`` (lss_fpn.py:459,493-494)."""
    if not imgs.is_cuda:
        raise RuntimeError("sgv3d_amd runs on the MI355X only (no CPU fallback)")
    bb = model.backbone
    with deferred_counters():                   # the BatchNorm step counters: one launch for all of them
        if isinstance(bb, bsm_lss_fpn.BSMLSSFPN):
            bev, aux = bsm_lss_fpn_forward(bb, imgs, mats_dict)
        else:
            bev, feats = lss_fpn_forward(bb, imgs, mats_dict, want_feats=True)
            aux = None
            if bb.is_train_height:
                assist = _nchw(conv(bb.assist_layer, feats))
                aux = (assist, assist)
        preds = head_forward(model.head, bev)
    if model.is_train_height:
        if not bb.is_train_height:
            raise RuntimeError("is_train_height: the backbone was built without it (backbone_conf['is_train_height'], "
                               "exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:236)")
        return preds, aux
    return preds
