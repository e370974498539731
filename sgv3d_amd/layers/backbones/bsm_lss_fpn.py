"""``BSMLSSFPN`` — the SGV3D background-suppressed view transform (BASELINE cfg-5) on MI355X.

Mirror of layers/backbones/bsm_lss_fpn.py: ``BSMLSSFPN(x_bound, ..., is_train_height, is_bsm)``
(:322-371), two image necks (``img_neck_16`` at stride 16, ``img_neck_8`` at stride 8 with
``upsample_strides=[0.5, 1, 2, 4]``, :364-370), the multi-scale ``MSCThead`` (:214-320) built from
``TaskHead`` (:178-201), ``TaskFPN`` (:203-212) and ``SABlock`` (:151-160), frustum at
``downsample_factor // 2`` (:343), and ``_forward_single_sweep`` (:485-559): softmax over the height
bins, 7-class semantic softmax, ``cat(context, semantic)`` (80 + 7 = 87 channels) zeroed where the
background probability exceeds 0.45, lift, voxel pooling.  Parameter names are the reference's
(``img_neck_16.*``, ``img_neck_8.*``, ``height_net.{reduce_conv0,...,context_conv1}.*``).

HIP specifics: the 87-channel transferred feature is carried as 88 channels (one zero channel) so that
every pixel row is 16-byte aligned for the lift / voxel-pooling / conv kernels; the extra channel
is dropped when the BEV map is handed out in the reference's NCHW layout.
"""
import copy

import torch
from torch import nn

from ... import hip_ops
from ...calibration import CalibrationCache
from ..blocks import BasicBlock, HipModule, build_backbone, build_neck, conv_bn
from .lss_fpn import ASPP, CACHE_CAMERA_GATES, FUSE_LIFT_SPLAT, HeightNet, LSSFPN, Mlp, SELayer, _require_hip_inference

__all__ = ['BSMLSSFPN']


class SABlock(HipModule):
    """Spatial attention block, bsm_lss_fpn.py:151-160: conv(x) * sigmoid(attention_conv(y))."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.attention = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False), nn.Sigmoid())
        self.conv = nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False)

    def hip_compile(self, device):
        return dict(att=conv_bn(self.attention[0], None, False, device), conv=conv_bn(self.conv, None, False, device))

    def hip_forward_residual(self, x, y, residual):
        """residual + conv(x) * sigmoid(attention(y))  (the form TaskFPN uses, :211)."""
        s = self.hip_state(x.device)
        dt = residual.dtype
        return hip_ops.add_mul_sigmoid(residual, s['conv'](x, out_dtype=dt), s['att'](y, out_dtype=dt))


class TaskHead(HipModule):
    """bsm_lss_fpn.py:178-201"""

    def __init__(self, in_channels, mid_channels, out_channels, with_head=True):
        super().__init__()
        self.with_head = with_head
        self.in_channels = in_channels
        self.mid_channels = mid_channels
        self.decoder = nn.Sequential(
            BasicBlock(mid_channels, mid_channels),
            BasicBlock(mid_channels, mid_channels),
            nn.Conv2d(mid_channels, mid_channels, 3, 1, 1),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True)
        )
        if self.with_head:
            self.head = nn.Conv2d(mid_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def hip_compile(self, device):
        s = dict(conv=conv_bn(self.decoder[2], self.decoder[3], True, device))
        if self.with_head:
            s['head'] = conv_bn(self.head, None, False, device)
        return s

    def hip_decoder(self, x):
        s = self.hip_state(x.device)
        dt = x.dtype if x.dtype == torch.bfloat16 else None       # bf16-activation mode: stay in the dtype of the input map
        x = self.decoder[0].hip_forward(x, dt)
        x = self.decoder[1].hip_forward(x, dt)
        return s['conv'](x, out_dtype=dt)

    def hip_head(self, feat, out=None, y_coff=0):
        return self.hip_state(feat.device)['head'](feat, out, y_coff=y_coff)


class TaskFPN(HipModule):
    """bsm_lss_fpn.py:203-212"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.reduce_conv = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.self_attention = SABlock(out_channels, out_channels)

    def hip_compile(self, device):
        return dict(reduce=conv_bn(self.reduce_conv, None, False, device))

    def hip_forward(self, feat0, feat1):
        s = self.hip_state(feat0.device)
        up = hip_ops.upsample_bilinear2x(feat0)                         # F.interpolate(scale_factor=2, bilinear), :210
        feat0 = s['reduce'](up, out_dtype=up.dtype)
        return self.self_attention.hip_forward_residual(feat1, feat0, feat0)   # feat0 + SA(feat1, feat0), :211


class MSCThead(HipModule):
    """bsm_lss_fpn.py:214-320"""

    def __init__(self, in_channels=[512, 512], mid_channels=[512, 256], depth_channels=90, semantic_channels=2,
                 context_channels=80):
        super().__init__()
        self.reduce_conv0 = nn.Sequential(
            nn.Conv2d(in_channels[0], mid_channels[0], kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(mid_channels[0]), nn.ReLU(inplace=True))
        self.reduce_conv1 = nn.Sequential(
            nn.Conv2d(in_channels[1], mid_channels[1], kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(mid_channels[1]), nn.ReLU(inplace=True))
        self.bn = nn.BatchNorm1d(27)
        self.scale0_mlp = Mlp(27, mid_channels[0], mid_channels[0])
        self.scale1_mlp = Mlp(27, mid_channels[1], mid_channels[1])
        self.scale0_se = SELayer(mid_channels[0])
        self.scale1_se = SELayer(mid_channels[1])
        self.aspp = ASPP(mid_channels[0], mid_channels[0])
        # stage one
        self.depth_head0 = TaskHead(mid_channels[0], mid_channels[0], depth_channels, with_head=False)
        self.semantic_head0 = TaskHead(mid_channels[0], mid_channels[0], semantic_channels)
        self.context_conv0 = nn.Sequential(
            nn.Conv2d(mid_channels[0], mid_channels[0], kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(mid_channels[0]),
            nn.ReLU(inplace=True)
        )
        self.depth_fpn = TaskFPN(mid_channels[0], mid_channels[1])
        self.semantic_fpn = TaskFPN(mid_channels[0], mid_channels[1])
        self.context_fpn = TaskFPN(mid_channels[0], mid_channels[1])
        # stage two
        self.depth_head1 = TaskHead(mid_channels[1], mid_channels[1], depth_channels)
        self.semantic_head1 = TaskHead(mid_channels[1], mid_channels[1], semantic_channels)
        self.context_conv1 = nn.Sequential(
            nn.Conv2d(mid_channels[1], mid_channels[1], kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(mid_channels[1]),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels[1], context_channels, kernel_size=1, stride=1, padding=0)
        )
        self.mid_channels = list(mid_channels)
        self.depth_channels = depth_channels
        self.semantic_channels = semantic_channels
        self.context_channels = context_channels

    def hip_compile(self, device):
        f = lambda t: t.detach().to(device).float().contiguous()
        bn = self.bn
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        bn_scale = f(bn.weight.float() * inv)
        bn_shift = f(bn.bias.float() - bn.running_mean.float() * bn.weight.float() * inv)
        s = dict(reduce0=conv_bn(self.reduce_conv0[0], self.reduce_conv0[1], True, device),
                 reduce1=conv_bn(self.reduce_conv1[0], self.reduce_conv1[1], True, device),
                 ctx0=conv_bn(self.context_conv0[0], self.context_conv0[1], True, device),
                 ctx1a=conv_bn(self.context_conv1[0], self.context_conv1[1], True, device),
                 ctx1b=conv_bn(self.context_conv1[3], None, False, device))
        for i in (0, 1):
            mlp, se = getattr(self, f'scale{i}_mlp'), getattr(self, f'scale{i}_se')
            c = self.mid_channels[i]
            w1, b1 = f(mlp.fc1.weight), f(mlp.fc1.bias)
            s[f'gate{i}'] = [
                ((w1 * bn_scale[None, :]).contiguous(), (b1 + w1 @ bn_shift).contiguous(), hip_ops.ACT_RELU),   # BN1d folded
                (f(mlp.fc2.weight), f(mlp.fc2.bias), hip_ops.ACT_NONE),
                (f(se.conv_reduce.weight.reshape(c, -1)), f(se.conv_reduce.bias), hip_ops.ACT_RELU),
                (f(se.conv_expand.weight.reshape(c, -1)), f(se.conv_expand.bias), hip_ops.ACT_SIGMOID),
            ]
        return s

    def camera_gates(self, mats_dict, device, out=None, tmp=None, run=None):
        """The SE gate vectors of the two scales (:262-299): a function of the calibration and the weights alone, kept per
        calibration by ``LSSFPN.calibration`` (layers/backbones/lss_fpn.py here).  ``out`` / ``tmp`` / ``run``: in place into the
        persistent vectors, skipped on the device when the flag says the calibration is the last one's (HeightNet.camera_gates)."""
        s = self.hip_state(device)
        v = HeightNet.mlp_input(mats_dict)                                   # :262-292
        res = []
        for i in (0, 1):
            layers = s[f'gate{i}']
            h = v
            if tmp is None:
                for w, b, act in layers:
                    h = hip_ops.dense(h, w, None, b, act)
            else:
                bufs = tmp.get(i)
                if bufs is None or bufs[0].shape[0] != v.shape[0]:
                    bufs = tmp[i] = [torch.zeros(v.shape[0], int(w.shape[0]), dtype=torch.float32, device=v.device) for w, _, _ in layers[:-1]]
                last = out[i] if out is not None else torch.zeros(v.shape[0], int(layers[-1][0].shape[0]), dtype=torch.float32, device=v.device)
                for (w, b, act), dst in zip(layers, bufs + [last]):
                    h = hip_ops.dense(h, w, None, b, act, out=dst, run=run)
            res.append(h)
        return res

    def hip_forward(self, feats, mats_dict, out_ld, gates=None):
        """feats = [stride-16 map, stride-8 map] NHWC.  Returns (height_context, semantic1, semantic0):
        height_context NHWC [B*N, H8, W8, out_ld] holds depth logits at [0, D) and the context at
        [D, D+80) (the caller composes the semantic part), semantic logits NHWC [.., 7].  ``gates``: ``camera_gates`` of this
        calibration when the caller keeps them; None: computed here."""
        s = self.hip_state(feats[0].device)
        if gates is None:
            gates = self.camera_gates(mats_dict, feats[0].device)
        gate = lambda i: gates[i]
        # bf16 mode: every mid-channel map of this head lives in HBM as bf16; the logits / context it returns are f32
        dt = hip_ops.activation_dtype(*[c.cout for c in (s['reduce0'], s['reduce1'], s['ctx0'], s['ctx1a'])])
        dev = feats[0].device
        # the two scales are independent up to the FPN stage, and so are the gate MLPs: branches of the captured graph
        # (hip_ops.run_parallel; in sequence outside a capture)

        def scale0_branch():
            r = s['reduce0'](feats[0], out_dtype=dt)
            return self.aspp.hip_forward(hip_ops.scale_channels(r, gate(0)))  # :300-306

        def scale1_branch():
            r = s['reduce1'](feats[1], out_dtype=dt)
            return hip_ops.scale_channels(r, gate(1))
        scale0, scale1 = hip_ops.run_parallel(dev, (scale0_branch, scale1_branch))
        B, H, W, _ = scale1.shape
        out = torch.empty(B, H, W, out_ld, dtype=torch.float32, device=scale1.device)
        # TaskHead(with_head=False).forward(feat) returns ``feat`` unchanged (:195-199): the decoder of
        # depth_head0 is never run by the reference, its parameters are dead weights.
        # The three tasks (:308-319) share only scale0 / scale1: three branches, two of them writing channel slices of `out`.

        def depth_task():
            depth_feat = self.depth_fpn.hip_forward(scale0, scale1)           # :308, :313
            self.depth_head1.hip_head(self.depth_head1.hip_decoder(depth_feat), out, y_coff=0)          # :317

        def semantic_task():
            semantic_feat = self.semantic_head0.hip_decoder(scale0)           # :309
            sem0 = self.semantic_head0.hip_head(semantic_feat)
            semantic_feat = self.semantic_fpn.hip_forward(semantic_feat, scale1)                        # :314
            return self.semantic_head1.hip_head(self.semantic_head1.hip_decoder(semantic_feat)), sem0   # :318

        def context_task():
            context_feat = s['ctx0'](scale0, out_dtype=dt)                    # :310
            context_feat = self.context_fpn.hip_forward(context_feat, scale1)                           # :315
            s['ctx1b'](s['ctx1a'](context_feat, out_dtype=dt), out, y_coff=self.depth_channels)         # :319
        _, (semantic1, semantic0), _ = hip_ops.run_parallel(dev, (depth_task, semantic_task, context_task))
        return out, semantic1, semantic0


class BSMLSSFPN(LSSFPN):
    def __init__(self, x_bound, y_bound, z_bound, d_bound, final_dim, output_channels, downsample_factor,
                 img_backbone_conf, img_neck_conf, height_net_conf, is_train_height, is_bsm):
        HipModule.__init__(self)
        import numpy as np
        self.downsample_factor = downsample_factor // 2                      # :343
        self.d_bound = d_bound
        self.final_dim = final_dim
        self.output_channels = output_channels
        self.is_train_height = is_train_height
        self.register_buffer('voxel_size', torch.Tensor([row[2] for row in [x_bound, y_bound, z_bound]]))
        self.register_buffer('voxel_coord',
                             torch.Tensor([row[0] + row[2] / 2.0 for row in [x_bound, y_bound, z_bound]]))
        nums = [(row[1] - row[0]) / row[2] for row in [x_bound, y_bound, z_bound]]
        for q in nums:
            assert abs(q - round(q)) < 1e-6, f"voxel bound does not divide evenly: {q}"
        self.register_buffer('voxel_num', torch.LongTensor([int(round(q)) for q in nums]))
        self.register_buffer('frustum', self.create_frustum())
        self.height_channels, _, _, _ = self.frustum.shape
        self.img_backbone = build_backbone(img_backbone_conf)
        self.img_backbone.init_weights()
        self.img_neck_16 = build_neck(img_neck_conf)
        self.img_neck_16.init_weights()
        neck8 = copy.deepcopy(dict(img_neck_conf))       # the reference mutates the caller's dict (:368)
        neck8['upsample_strides'] = [0.5, 1, 2, 4]
        self.img_neck_8 = build_neck(neck8)
        self.img_neck_8.init_weights()
        self.height_net = self._configure_height_net(height_net_conf)
        self._voxel_num_host = tuple(int(round(q)) for q in nums)
        self._voxel_coord_host = [float(np.float32(row[0] + row[2] / 2.0)) for row in [x_bound, y_bound, z_bound]]
        self._voxel_size_host = [float(np.float32(row[2])) for row in [x_bound, y_bound, z_bound]]
        self.semantic_channels = height_net_conf['semantic_channels']
        self.background_threshold = 0.45                                     # :528
        self.fuse_lift_splat = FUSE_LIFT_SPLAT      # as LSSFPN
        self.calib_cache = CalibrationCache()

    def _configure_height_net(self, height_net_conf):
        return MSCThead(
            in_channels=height_net_conf['in_channels'],
            mid_channels=height_net_conf['mid_channels'],
            depth_channels=self.height_channels,
            semantic_channels=height_net_conf['semantic_channels'],
            context_channels=self.output_channels,
        )

    @property
    def bev_channels(self):
        return self.output_channels + self.semantic_channels                  # 80 + 7

    def get_cam_feats_nhwc(self, imgs):
        """get_cam_feats (bsm_lss_fpn.py:462-479): one backbone pass, two necks."""
        batch_size, num_sweeps, num_cams, num_channels, imH, imW = imgs.shape
        imgs = imgs.reshape(batch_size * num_sweeps * num_cams, num_channels, imH, imW).float().contiguous()
        cin_pad = self.img_backbone.hip_state(imgs.device)['cin_pad']
        feats = self.img_backbone.hip_forward(hip_ops.nchw_to_nhwc(imgs, c_pad=cin_pad), split_tag="img_backbone.stage1")
        return [n.hip_forward(feats, out_dtype=hip_ops.activation_dtype(*n.out_channels)) for n in (self.img_neck_16, self.img_neck_8)]

    def _forward_single_sweep(self, sweep_index, sweep_imgs, mats_dict, nhwc_out=False):
        """bsm_lss_fpn.py:485-559"""
        batch_size, num_sweeps, num_cams, num_channels, img_height, img_width = sweep_imgs.shape
        img_feats = self.get_cam_feats_nhwc(sweep_imgs)
        D, C = self.height_channels, self.bev_channels
        Cp = (C + 3) // 4 * 4                                                 # 87 -> 88 (zero channel)
        geom_xyz, plan = self.calibration(mats_dict, sweep_index)             # :540-553 (cached per calibration, with the gates)
        gates = self.calib_cache.entry(0).gates if CACHE_CAMERA_GATES else None
        hc, semantic1, _semantic0 = self.height_net.hip_forward(img_feats, mats_dict, out_ld=D + Cp, gates=gates)
        hip_ops.bsm_compose(hc, semantic1, D, self.output_channels, self.semantic_channels,
                            self.background_threshold)                        # :521-529
        fH, fW = int(hc.shape[1]), int(hc.shape[2])
        if self.fuse_lift_splat and num_cams == 1:
            prob, _ = hip_ops.lift(hc, D, Cp, want_prob=True, want_lifted=False)
            ldo = hip_ops.pad_channels(Cp) if (nhwc_out and hip_ops.activation_dtype(Cp) == torch.bfloat16 and getattr(self, '_single_sweep', True)) else 0
            ctx = torch.empty(batch_size, fH, fW, Cp, dtype=torch.float32, device=hc.device)     # (f32 rows, as LSSFPN)
            hip_ops.copy_channels(hc, ctx, coff=D)
            bev = plan.lift_splat(prob, ctx.view(batch_size, fH * fW, Cp), out_bf16_ld=ldo)
        else:
            _, lifted = hip_ops.lift(hc, D, Cp, lifted_dtype=hip_ops.activation_dtype(Cp))
            ldo = hip_ops.pad_channels(Cp) if (nhwc_out and lifted.dtype == torch.bfloat16 and getattr(self, '_single_sweep', True)) else 0    # as in LSSFPN
            bev = plan.pool(lifted.view(batch_size, num_cams * D * fH * fW, Cp), out_bf16_ld=ldo)     # voxel_pooling of :554-555
        feature_map = bev.permute(0, 3, 1, 2)
        nhwc = feature_map.permute(0, 2, 3, 1)                                # [B, Y, X, 88]
        if nhwc_out:
            return nhwc
        return hip_ops.nhwc_to_nchw(nhwc, channels=C)                         # the reference's [B, 87, Y, X]

    def forward(self, sweep_imgs, mats_dict, timestamps=None, nhwc_out=False):
        _require_hip_inference(self, sweep_imgs)
        return LSSFPN.forward(self, sweep_imgs, mats_dict, timestamps, nhwc_out=nhwc_out)
