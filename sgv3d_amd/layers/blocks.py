"""Building blocks the reference pulls from third-party packages, re-created here as parameter
containers with the SAME constructor configs and state_dict names, plus a HIP forward.

Reference call sites: ``build_backbone(img_backbone_conf)`` / ``build_neck(img_neck_conf)``
(layers/backbones/lss_fpn.py:296-297), ``BasicBlock`` (lss_fpn.py:186-188), ``build_backbone(
bev_backbone_conf)`` / ``build_neck(bev_neck_conf)`` (layers/heads/bev_height_head.py:75-78).
Upstream definitions followed: mmdet 2.19.0 ``ResNet`` / ``BasicBlock`` / ``Bottleneck``
(style='pytorch'), mmdet3d 0.18.1 ``SECONDFPN``.

The ``torch.nn`` leaf modules (Conv2d, BatchNorm2d, ...) are used only as parameter/buffer holders so
that Lightning checkpoints load by name; arithmetic runs in ``hip_forward`` through the C ABI on
NHWC float32 buffers.  BatchNorm is the eval-mode affine (running statistics) folded into the
convolution epilogue.
"""
import torch
from torch import nn

from .. import hip_ops
from ..hip_ops import PackedConv, fold_bn


def _kaiming_(m):
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class HipModule(nn.Module):
    """nn.Module whose packed HIP state (repacked weights, folded BN) is built lazily per device."""

    _GENERATION = 0          # process-wide count of hip_compile runs: a compiled state's identity that is never reused

    def __init__(self):
        super().__init__()
        self._hip = None
        self._hip_gen = 0

    def hip_state(self, device):
        if self._hip is None or self._hip[0] != device:
            self._hip = (device, self.hip_compile(device))
            HipModule._GENERATION += 1
            self._hip_gen = HipModule._GENERATION
        return self._hip[1]

    def hip_generation(self, device):
        """Identity of the packed state on ``device`` (compiling it if needed): a monotonically increasing number, unlike ``id()`` of
        the state object, whose address CPython hands to the next allocation after ``refresh()`` dropped it -- anything cached beside
        the packed weights (the height net's camera gates) is keyed on this."""
        self.hip_state(device)
        return self._hip_gen

    def hip_invalidate(self):
        self._hip = None
        for m in self.children():
            if isinstance(m, HipModule):
                m.hip_invalidate()
            else:
                for sub in m.modules():
                    if isinstance(sub, HipModule):
                        sub.hip_invalidate()

    def hip_compile(self, device):  # pragma: no cover - abstract
        raise NotImplementedError


def conv_bn(conv, bn=None, relu=False, device=None, cin_pad=None, pad_out=False):
    """PackedConv of an nn.Conv2d (+ optional eval BatchNorm2d folded, + ReLU)."""
    assert conv.groups == 1
    k = conv.kernel_size[0]
    assert conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
    if bn is not None:
        scale, shift = fold_bn(bn, conv.bias)
    else:
        scale, shift = None, (conv.bias.detach() if conv.bias is not None else None)
    return PackedConv(conv.weight, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0],
                      scale=scale, shift=shift, relu=relu, device=device, cin_pad=cin_pad, pad_out=pad_out)


# ------------------------------------------------------------------------------------------------
# mmdet ResNet family
# ------------------------------------------------------------------------------------------------
class BasicBlock(HipModule):
    """mmdet.models.backbones.resnet.BasicBlock: conv3x3-BN-ReLU-conv3x3-BN, + identity, ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def hip_compile(self, device):
        # channel counts are padded to the owner network's alignment (ResNet.hip_compile hands it down; 4, or 32 in
        # bf16-activation mode): this block's input was padded the same way by its producer
        al = getattr(self, '_hip_align', None) or hip_ops.channel_align()
        pc = lambda c: hip_ops.pad_channels(c, al)
        s = dict(c1=conv_bn(self.conv1, self.bn1, True, device, cin_pad=pc(self.conv1.in_channels), pad_out=al),
                 c2=conv_bn(self.conv2, self.bn2, True, device, cin_pad=pc(self.conv2.in_channels), pad_out=al))
        if self.downsample is not None:
            s['ds'] = conv_bn(self.downsample[0], self.downsample[1], False, device,
                              cin_pad=pc(self.downsample[0].in_channels), pad_out=al)
        return s

    def hip_forward(self, x, act_dtype=None):
        s = self.hip_state(x.device)
        if 'ds' in s:      # the strided shortcut beside conv1 (two branches of the captured graph, hip_ops.run_parallel)
            out, identity = hip_ops.run_parallel(x.device, (lambda: s['c1'](x, out_dtype=act_dtype),
                                                            lambda: s['ds'](x, out_dtype=act_dtype)))
        else:
            identity, out = x, s['c1'](x, out_dtype=act_dtype)
        return s['c2'](out, residual=identity, out_dtype=act_dtype)      # relu(bn2(conv2) + identity)


class Bottleneck(HipModule):
    """mmdet Bottleneck, style='pytorch' (the stride sits on the 3x3)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def hip_compile(self, device):
        s = dict(c1=conv_bn(self.conv1, self.bn1, True, device), c2=conv_bn(self.conv2, self.bn2, True, device),
                 c3=conv_bn(self.conv3, self.bn3, True, device))
        if self.downsample is not None:
            s['ds'] = conv_bn(self.downsample[0], self.downsample[1], False, device)
        return s

    def hip_forward(self, x, act_dtype=None):
        s = self.hip_state(x.device)
        if 'ds' in s:
            # the shortcut convolution of a stage's first block beside conv1 -> conv2 (two branches of the captured graph)
            def main():
                o = s['c1'](x, out_dtype=act_dtype)
                if act_dtype == torch.bfloat16 and hip_ops.conv_pair_eligible(s['c2'], s['c3'], o):
                    return o, None                               # (the fused pair needs the residual: finished after the join)
                return o, s['c2'](o, out_dtype=act_dtype)
            (out, mid), identity = hip_ops.run_parallel(x.device, (main, lambda: s['ds'](x, out_dtype=act_dtype)))
            if mid is not None:
                return s['c3'](mid, residual=identity, out_dtype=act_dtype)
        else:
            identity, out = x, s['c1'](x, out_dtype=act_dtype)
        # bf16 configs, 256-channel bottlenecks (ResNet layer 3): conv2 + conv3 in one launch, the map between them in LDS
        if act_dtype == torch.bfloat16 and hip_ops.conv_pair_choice(s['c2'], s['c3'], out, identity):
            return hip_ops.conv_pair_bf16(s['c2'], s['c3'], out, identity)
        return s['c3'](s['c2'](out, out_dtype=act_dtype), residual=identity, out_dtype=act_dtype)


class ResNet(HipModule):
    """mmdet 2.19.0 ``ResNet`` (deep_stem=False, avg_down=False, style='pytorch', no DCN/plugins).

    Accepts the config dicts of the reference verbatim, e.g.
    ``dict(type='ResNet', depth=50, frozen_stages=0, out_indices=[0,1,2,3], norm_eval=False,
    init_cfg=...)`` (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:44-52) and
    ``dict(type='ResNet', in_channels=80, depth=18, num_stages=3, strides=(1,2,2), dilations=(1,1,1),
    out_indices=[0,1,2], norm_eval=False, base_channels=160)`` (:77-87).
    """
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, norm_eval=True, init_cfg=None,
                 pretrained=None, **unused):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert style == 'pytorch' and not deep_stem and not avg_down
        assert all(d == 1 for d in dilations[:num_stages])
        block, stage_blocks = self.arch_settings[depth]
        self.depth = depth
        self.deep_stem = deep_stem
        self.out_indices = list(out_indices)
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        stem_channels = stem_channels or base_channels
        self.in_channels = in_channels
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(stem_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        inplanes = stem_channels
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            layers = []
            for j in range(stage_blocks[i]):
                stride = strides[i] if j == 0 else 1
                downsample = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    downsample = nn.Sequential(
                        nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                        nn.BatchNorm2d(planes * block.expansion))
                layers.append(block(inplanes, planes, stride, downsample))
                inplanes = planes * block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._freeze_stages()

    @property
    def norm1(self):
        return self.bn1

    def _freeze_stages(self):
        """mmdet 2.19.0 ``ResNet._freeze_stages`` (the configs of the reference set ``frozen_stages=0`` on the image backbone,
        exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:48, exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:57):
        with ``frozen_stages >= 0`` the stem's BatchNorm stays in eval mode and ``conv1`` / ``bn1`` stop training; stages
        ``1 .. frozen_stages`` likewise.  Called from the constructor and re-applied by every ``train()``, as upstream."""
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        """mmdet ``ResNet.train``: the frozen stages are re-frozen, and with ``norm_eval`` every BatchNorm keeps its running
        statistics while the rest of the network trains."""
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def frozen_stem(self):
        """True when conv1 + bn1 are constants of the training step (``frozen_stages >= 0``): the training forward then runs the stem
        as the inference path does -- one convolution with the folded BatchNorm and ReLU in its epilogue, no statistics pass, no
        weight gradient."""
        return self.frozen_stages >= 0 and not self.bn1.training and not any(p.requires_grad for m in (self.conv1, self.bn1) for p in m.parameters())

    def init_weights(self):
        """Kaiming / constant init (mmdet's fallback when no checkpoint is given; the reference asks
        for torchvision weights, lss_fpn.py:299, which need network access and load by name)."""
        for m in self.modules():
            _kaiming_(m)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.zeros_(m.bn3.weight)
            elif isinstance(m, BasicBlock):
                nn.init.zeros_(m.bn2.weight)

    def hip_compile(self, device):
        align = hip_ops.channel_align()
        # a wide input (the 80 / 87-channel BEV map of the head's trunk) is padded like the activations: in bf16-activation
        # mode the producer (voxel pooling) writes rows of pad_channels(C) channels; an image (3 channels) just to 4
        cin_pad = hip_ops.pad_channels(self.in_channels, align) if self.in_channels >= 64 else (self.in_channels + 3) // 4 * 4
        for m in self.modules():
            if isinstance(m, BasicBlock):
                m._hip_align = align          # the blocks compile lazily: same padding as the stem, whatever the mode is then
        return dict(stem=conv_bn(self.conv1, self.bn1, True, device, cin_pad=cin_pad, pad_out=align), cin_pad=cin_pad, align=align)

    def hip_stem(self, x_nhwc):
        """conv1 + bn1 + relu on an NHWC input whose channels are already padded to a multiple of 4."""
        return self.hip_state(x_nhwc.device)['stem'](x_nhwc)

    def act_dtype(self):
        """bf16 mode keeps this network's activations as bf16 tensors in HBM (hip_ops.BF16_ACTIVATIONS) when every
        channel count is a multiple of 8 (16-byte rows); None = fp32 tensors."""
        if not (hip_ops.MFMA_BF16 and hip_ops.BF16_ACTIVATIONS) or hip_ops.MFMA_F32X3:
            return None
        align = self._hip[1].get('align', 4) if self._hip is not None else hip_ops.channel_align()   # as compiled
        basic = any(isinstance(m, BasicBlock) for m in self.modules())
        pad = (lambda c: hip_ops.pad_channels(c, align)) if basic else (lambda c: c)    # as compiled (Bottlenecks are not padded)
        chans = [pad(self.conv1.out_channels)] + [pad(m.out_channels) for m in self.modules() if isinstance(m, nn.Conv2d)]
        return torch.bfloat16 if all(c % 8 == 0 for c in chans) else None

    def hip_forward(self, x_nhwc, use_maxpool=True, split_tag=None):
        """``split_tag``: the image backbone names a cut point after its first stage (hip_ops.graph_split_point: a capture
        that replays the frame as two graphs launches the short one first, so the GPU starts while the host is still
        submitting the long one)."""
        dt = self.act_dtype()
        x = self.hip_state(x_nhwc.device)['stem'](x_nhwc, out_dtype=dt)
        if use_maxpool:
            x = hip_ops.maxpool3x3s2(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            for blk in getattr(self, name):
                x = blk.hip_forward(x, dt)
            if i in self.out_indices:
                outs.append(x)
            if i == 0 and split_tag is not None:
                hip_ops.graph_split_point(split_tag)
        return outs


class SECONDFPN(HipModule):
    """mmdet3d 0.18.1 ``SECONDFPN``: per level ConvTranspose2d(k = s, stride s) for s >= 1, or
    Conv2d(k = 1/s, stride 1/s) for s < 1; BN(eps=1e-3, momentum=0.01); ReLU; concat on channels.
    Configs: exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:53-59 and :89-92."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 norm_cfg=None, upsample_cfg=None, conv_cfg=None, use_conv_for_no_stride=False, init_cfg=None,
                 **unused):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        norm_cfg = norm_cfg or dict(type='BN', eps=1e-3, momentum=0.01)
        self.in_channels = list(in_channels)
        self.out_channels = list(out_channels)
        self.upsample_strides = list(upsample_strides)
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                s = int(stride)
                layer = nn.ConvTranspose2d(in_channels[i], oc, kernel_size=s, stride=s, bias=False)
            else:
                s = int(round(1 / stride))
                layer = nn.Conv2d(in_channels[i], oc, kernel_size=s, stride=s, bias=False)
            bn = nn.BatchNorm2d(oc, eps=norm_cfg.get('eps', 1e-3), momentum=norm_cfg.get('momentum', 0.01))
            deblocks.append(nn.Sequential(layer, bn, nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)

    def init_weights(self):
        for m in self.modules():
            _kaiming_(m)

    def hip_compile(self, device):
        convs = []
        for blk in self.deblocks:
            layer, bn = blk[0], blk[1]
            scale, shift = fold_bn(bn)
            if isinstance(layer, nn.ConvTranspose2d) and layer.stride[0] == 1 and not hip_ops.MFMA_BF16:
                # ConvTranspose2d(k = 1, stride 1) IS a 1x1 convolution with the weight's first two axes swapped: as such it
                # gets the row-linear epilogue and the pointwise / five-per-CU kernels (the transposed-conv store decodes a tap
                # per element: 31 -> 15 us for the BEV neck's 80 -> 64 level at 256x256).  (bf16 mode keeps the transposed form
                # its committed per-layer choices were measured with.)
                convs.append(PackedConv(layer.weight.detach().permute(1, 0, 2, 3).contiguous(), scale=scale, shift=shift, relu=True,
                                        device=device))
            elif isinstance(layer, nn.ConvTranspose2d):
                convs.append(PackedConv(layer.weight, stride=layer.stride[0], transposed=True, scale=scale,
                                        shift=shift, relu=True, device=device))
            else:
                convs.append(PackedConv(layer.weight, stride=layer.stride[0], scale=scale, shift=shift, relu=True,
                                        device=device))
        return convs

    def hip_forward(self, feats, out=None, out_dtype=None):
        """feats: list of NHWC maps -> one NHWC map [B, H, W, sum(out_channels)] (levels written
        straight into their channel slice: no torch.cat).  ``out_dtype=torch.bfloat16`` (bf16-activation mode, every
        level's channel count a multiple of 8): the concatenated map is a bf16 tensor."""
        convs = self.hip_state(feats[0].device)
        B, h0, w0, _ = feats[0].shape
        oh, ow = convs[0].out_hw(int(h0), int(w0))
        total = sum(self.out_channels)
        if out is None:
            if out_dtype == torch.bfloat16 and any(c % 8 for c in self.out_channels):
                out_dtype = None
            out = torch.empty(B, oh, ow, total, dtype=out_dtype or torch.float32, device=feats[0].device)
        off, jobs = 0, []
        for conv, f, oc in zip(convs, feats, self.out_channels):
            assert conv.out_hw(int(f.shape[1]), int(f.shape[2])) == (oh, ow), \
                "SECONDFPN levels do not align (the reference's torch.cat would fail too)"
            jobs.append(lambda conv=conv, f=f, off=off: conv(f, out, y_coff=off))
            off += oc
        # the levels are independent (each writes its own channel slice) and none of them fills the chip: side by side in the
        # captured graph (hip_ops.run_parallel), the largest level first
        hip_ops.run_parallel(out.device, jobs)
        return out


_BACKBONES = {'ResNet': ResNet}
_NECKS = {'SECONDFPN': SECONDFPN}


def build_backbone(cfg):
    cfg = dict(cfg)
    return _BACKBONES[cfg.pop('type')](**cfg)


def build_neck(cfg):
    cfg = dict(cfg)
    return _NECKS[cfg.pop('type')](**cfg)
