"""``BEVHeightHead`` — CenterPoint head on the BEV map, MI355X forward.

Mirror of layers/heads/bev_height_head.py:31-111 (constructor keywords and defaults :13-64, forward
:85-111) on top of what the reference inherits from mmdet3d 0.18.1 ``CenterHead`` / ``SeparateHead``
(SURVEY.md §2.2): ``shared_conv`` = Conv3x3(in_channels -> 64, no bias) + BN + ReLU, then per task a
``SeparateHead`` whose branches ``reg, height, dim, rot, vel, heatmap`` are each
Conv3x3(64 -> 64, no bias) + BN + ReLU followed by Conv3x3(64 -> c, bias) (heatmap bias -2.19).
Parameter names: ``trunk.*``, ``neck.deblocks.*``, ``shared_conv.{conv,bn}``,
``task_heads.{t}.{branch}.0.{conv,bn}``, ``task_heads.{t}.{branch}.1``.

HIP forward ("CenterHead convs fused per scale"):
  trunk (7x7 stem without max-pool, 3 BasicBlock stages) -> SECONDFPN into one 256-ch map ->
  shared conv -> ONE 3x3 conv producing the 64-channel hidden maps of all 36 branches
  (64 -> 36*64, the 36 first-layer convs share their input) -> ONE kernel for the 36 final 3x3
  convs, writing a single NCHW [B, 70, H, W] buffer whose channel slices are the returned maps.
Return structure is the reference's: ``tuple(task -> [dict(branch -> Tensor[B, c, H, W])])``.

get_bboxes (decode + circle NMS) runs on the device (csrc/decode.hip).  get_targets / loss belong to
the training row of SURVEY.md §8(f) and are not implemented in this round (they raise).
"""
import torch
from torch import nn

from ... import hip_ops
from ...hip_ops import PackedConv, fold_bn
from ..blocks import HipModule, build_backbone, build_neck, conv_bn

__all__ = ['BEVHeightHead']

bev_backbone_conf = dict(
    type='ResNet',
    in_channels=80,
    depth=18,
    num_stages=3,
    strides=(1, 2, 2),
    dilations=(1, 1, 1),
    out_indices=[0, 1, 2],
    norm_eval=False,
    base_channels=160,
)

bev_neck_conf = dict(type='SECONDFPN',
                     in_channels=[160, 320, 640],
                     upsample_strides=[2, 4, 8],
                     out_channels=[64, 64, 128])


class ConvModule(nn.Module):
    """mmcv ConvModule(conv -> bn -> relu) parameter holder with mmcv's attribute names."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True)
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')


class SeparateHead(nn.Module):
    """mmdet3d 0.18.1 SeparateHead (parameter holder): one Sequential per branch."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19, **unused):
        super().__init__()
        self.heads = heads
        self.init_bias = init_bias
        for head in self.heads:
            classes, num_conv = self.heads[head]
            conv_layers = []
            c_in = in_channels
            for _ in range(num_conv - 1):
                conv_layers.append(ConvModule(c_in, head_conv, final_kernel, padding=final_kernel // 2))
                c_in = head_conv
            conv_layers.append(nn.Conv2d(head_conv, classes, final_kernel, stride=1, padding=final_kernel // 2, bias=True))
            self.__setattr__(head, nn.Sequential(*conv_layers))
        self.init_weights()

    def init_weights(self):
        for head in self.heads:
            if head == 'heatmap':
                self.__getattr__(head)[-1].bias.data.fill_(self.init_bias)


class BEVHeightHead(HipModule):
    """Head for BEVHeight (see module docstring).  Keyword arguments as the reference,
    layers/heads/bev_height_head.py:46-64."""

    def __init__(
        self,
        in_channels=256,
        tasks=None,
        bbox_coder=None,
        common_heads=dict(),
        loss_cls=dict(type='GaussianFocalLoss', reduction='mean'),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        gaussian_overlap=0.1,
        min_radius=2,
        train_cfg=None,
        test_cfg=None,
        bev_backbone_conf=bev_backbone_conf,
        bev_neck_conf=bev_neck_conf,
        separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
        share_conv_channel=64,
        num_heatmap_convs=2,
    ):
        super().__init__()
        num_classes = [len(t['class_names']) for t in tasks]
        self.class_names = [t['class_names'] for t in tasks]
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.bbox_coder_cfg = bbox_coder
        self.loss_cls_cfg, self.loss_bbox_cfg = loss_cls, loss_bbox
        self.norm_bbox = True
        self.shared_conv = ConvModule(in_channels, share_conv_channel, 3, padding=1)
        self.task_heads = nn.ModuleList()
        for num_cls in num_classes:
            heads = dict(common_heads)
            heads.update(dict(heatmap=(num_cls, num_heatmap_convs)))
            sh = dict(separate_head)
            sh.pop('type', None)
            self.task_heads.append(SeparateHead(in_channels=share_conv_channel, heads=heads, head_conv=64, **sh))
        self.trunk = build_backbone(bev_backbone_conf)
        self.trunk.init_weights()
        self.neck = build_neck(bev_neck_conf)
        self.neck.init_weights()
        del self.trunk.maxpool                                   # bev_height_head.py:79
        self.gaussian_overlap = gaussian_overlap
        self.min_radius = min_radius
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.share_conv_channel = share_conv_channel

    # ---------------------------------------------------------------------------------------------
    def _branches(self):
        """[(task index, branch name, Sequential)] in the reference's dict order."""
        out = []
        for t, th in enumerate(self.task_heads):
            for name in th.heads:
                out.append((t, name, getattr(th, name)))
        return out

    def hip_compile(self, device):
        br = self._branches()
        hc = 64
        for _, name, seq in br:
            assert len(seq) == 2 and seq[0].conv.out_channels == hc, "fused head expects (ConvModule, Conv2d) branches"
            assert seq[1].out_channels <= 4
        # first layers: one conv 64 -> nb*64 with per-channel folded BN
        w1 = torch.cat([seq[0].conv.weight.detach() for _, _, seq in br], 0)
        sc, sh = zip(*[fold_bn(seq[0].bn, seq[0].conv.bias) for _, _, seq in br])
        first = PackedConv(w1, pad=1, scale=torch.cat(sc), shift=torch.cat(sh), relu=True, device=device)
        # final layers: [sum_c, 3, 3, 64] + bias + branch map
        w2 = torch.cat([seq[1].weight.detach().permute(0, 2, 3, 1) for _, _, seq in br], 0)
        b2 = torch.cat([seq[1].bias.detach() for _, _, seq in br], 0)
        branch_of_out, out_begin, slices, off = [], [0], [], 0
        for i, (t, name, seq) in enumerate(br):
            c = seq[1].out_channels
            branch_of_out += [i] * c
            slices.append((t, name, off, c))
            off += c
            out_begin.append(off)
        f = lambda t_: t_.to(device).float().contiguous()
        ob_dev = torch.tensor(out_begin, dtype=torch.int32, device=device)
        # bf16 mode: fragment-ordered bf16 copies of both branch layers for the fused bf16 head kernel (csrc/head_bf16.hip)
        w1_bf16 = None
        if hip_ops.MFMA_BF16 and hip_ops.FUSED_HEAD and tuple(w1.shape[1:]) == (64, 3, 3):
            w1_bf16 = hip_ops.pack_centerhead_bf16(w1.to(device), w2.to(device), ob_dev)
        # fp32 mode: the first layers as the F(4x4) fragment stream of the fused kernel of csrc/head_wino4.hip
        u_f4 = None
        if hip_ops.FUSED_HEAD and not hip_ops.MFMA_BF16 and tuple(w1.shape[1:]) == (64, 3, 3) and hc == 64:
            u_f4 = hip_ops.pack_centerhead_f4(w1.to(device).float())
        return dict(w1_bf16=w1_bf16, u_f4=u_f4, shared=conv_bn(self.shared_conv.conv, self.shared_conv.bn, True, device), first=first,
                    w2=f(w2), b2=f(b2), branch_of_out=torch.tensor(branch_of_out, dtype=torch.int32, device=device),
                    out_begin=ob_dev,
                    slices=slices, nb=len(br), hc=hc, total=off)

    def hip_forward(self, x):
        """x: BEV map NHWC [B, Y, X, C] -> the reference's nested prediction structure."""
        s = self.hip_state(x.device)
        trunk_outs = [x]                                               # bev_height_head.py:97
        dt = self.trunk.act_dtype()                                    # bf16 mode: bf16 tensors between the trunk's layers
        h = self.trunk.hip_state(x.device)['stem'](x, out_dtype=dt)    # conv1 + norm1 + relu, no maxpool (:101-103)
        for i, layer_name in enumerate(self.trunk.res_layers):         # :104-108
            for blk in getattr(self.trunk, layer_name):
                h = blk.hip_forward(h, dt)
            if i in self.trunk.out_indices:
                trunk_outs.append(h)
        # bf16-activation mode with the fused bf16 head: neck output and shared map stay bf16 tensors as well
        nd = dt if (dt is not None and hip_ops.MFMA_BF16 and s.get('w1_bf16') is not None) else None
        fpn_output = self.neck.hip_forward(trunk_outs, out_dtype=nd)   # :109
        shared = s['shared'](fpn_output, out_dtype=nd if fpn_output.dtype == torch.bfloat16 else None)   # CenterHead.forward_single
        if hip_ops.MFMA_BF16 and s.get('w1_bf16') is not None:
            # bf16 matrix cores: both branch layers in one kernel, hidden maps in LDS as bf16
            out = hip_ops.centerhead_branches_bf16(shared, s['w1_bf16'], s['first'].scale, s['first'].shift, s['b2'],
                                                   s['out_begin'], s['nb'])
        elif hip_ops.FUSED_HEAD and not hip_ops.MFMA_BF16 and s['first'].wino_ok and s['first'].cin <= 64 and s['hc'] == 64:
            path = self._branch_path(s, shared)
            if path == 2:
                # both branch layers in one kernel, first layers in F(4x4) form: the hidden maps never leave the registers
                out = self._fused_f4(s, shared)
            elif path == 0:
                # the F(2x2) fused kernel: the [nb,B,H,W,64] hidden maps stay in LDS
                out = hip_ops.centerhead_branches(shared, s['first'], s['w2'], s['b2'], s['out_begin'], s['nb'])
            else:
                out = self._two_kernel_branches(s, shared)
        else:
            out = self._two_kernel_branches(s, shared)
        ret = [dict() for _ in self.task_heads]
        for t, name, off, c in s['slices']:
            ret[t][name] = out[:, off:off + c]                         # [B, c, H, W] views of one buffer
        return tuple([d] for d in ret)                                 # multi_apply over one level

    @staticmethod
    def _two_kernel_branches(s, shared):
        """First layers of all branches as ONE convolution (64 -> nb x 64, hidden maps [nb,B,H,W,64] in HBM; its algorithm is
        chosen per load like any layer's -- with three frames in flight the three-launch F(4x4) Winograd), then the final
        3x3 convolutions of all branches in one launch."""
        hidden = s['first'](shared, group_planes=s['hc'])
        return hip_ops.head_final_conv(hidden, s['w2'], s['b2'], s['branch_of_out'], s['nb'], s['hc'])

    @staticmethod
    def _fused_f4(s, shared):
        return hip_ops.centerhead_branches_f4(shared, s['u_f4'], s['first'].scale, s['first'].shift, s['w2'], s['b2'],
                                              s['out_begin'], s['nb'])

    def _branch_path(self, s, shared):
        """2: the fused F(4x4) kernel (both branch layers, transformed input resident in LDS, hidden maps in registers; 0.53 ms
        at 256x256); 0: the fused F(2x2) kernel (hidden maps in LDS; 0.98 ms); 1: the two-kernel path (hidden maps through
        HBM).  ``hip_ops.HEAD_PATH`` names one; with "auto" all available ones are timed under the load they will run in
        (hip_ops.TUNE_STREAMS concurrent copies) at the first call outside a graph capture and the choice is kept in
        hip_ops.TUNE_DB."""
        f4_ok = (s.get('u_f4') is not None and shared.dtype == torch.float32 and int(shared.shape[-1]) >= 64
                 and s['first'].cin == 64)
        want = hip_ops.HEAD_PATH
        if want in (0, 1, 2):                                # an explicit setting wins over any recorded measurement
            return want if (want != 2 or f4_ok) else 0
        B, H, W, _ = (int(v) for v in shared.shape)
        sig = f"centerhead_branches3|{B}x{H}x{W}x{s['nb']}|ts{hip_ops.TUNE_STREAMS}"
        ok = {0: True, 1: s['first'].wino4_ok(), 2: f4_ok}
        hit = hip_ops.TUNE_DB.get(sig) if hip_ops.AUTOTUNE else None
        if hit is not None and ok.get(int(hit[0]) - 100, False):
            return int(hit[0]) - 100
        if not hip_ops.AUTOTUNE or torch.cuda.is_current_stream_capturing():
            return 2 if f4_ok else 0
        cands = {0: lambda: hip_ops.centerhead_branches(shared, s['first'], s['w2'], s['b2'], s['out_begin'], s['nb']),
                 1: lambda: self._two_kernel_branches(s, shared), 2: lambda: self._fused_f4(s, shared)}
        cands = {k: f for k, f in cands.items() if ok[k]}
        for f in cands.values():
            f()                                              # warm (the first layer measures its own candidates here)
        torch.cuda.synchronize(shared.device)
        times = {k: hip_ops.time_callable(f, shared.device, rounds=2) for k, f in cands.items()}
        choice = min(times, key=times.get)
        hip_ops.TUNE_DB[sig] = (100 + choice, 1)
        return choice

    def forward(self, x, nhwc=False):
        """x: [B, C, Y, X] (reference layout) or, with ``nhwc``, the NHWC buffer [B, Y, X, C]."""
        if not x.is_cuda:
            raise RuntimeError("sgv3d_amd runs on the MI355X only (no CPU fallback)")
        if self.training:
            raise NotImplementedError("HIP path = inference forward; call model.eval()")
        cin_pad = self.trunk.hip_state(x.device)['cin_pad']            # channels the trunk's first convolution reads
        if not nhwc:
            x = hip_ops.nchw_to_nhwc(x.float().contiguous(), c_pad=max(cin_pad, (int(x.shape[1]) + 3) // 4 * 4))
        elif not x.is_contiguous():
            x = x.contiguous()
        if int(x.shape[-1]) < cin_pad:                                 # a producer that did not pad (fused lift-splat, external maps)
            x = torch.nn.functional.pad(x, (0, cin_pad - int(x.shape[-1])))
        return self.hip_forward(x)

    # --------------------------------------------------------------- training side
    def get_targets(self, gt_bboxes_3d, gt_labels_3d):
        """mmdet3d ``CenterHead.get_targets`` over ``get_targets_single`` (bev_height_head.py:113-253) as one
        device launch for the whole batch (the reference loops over boxes in Python with device scalars).

        ``gt_bboxes_3d``: list (per sample) of [N_i, 9] float tensors (x, y, z, w, l, h, yaw, vx, vy) or objects
        with a ``.tensor``; ``gt_labels_3d``: list of [N_i] integer tensors.  Returns the reference's tuple
        ``(heatmaps, anno_boxes, inds, masks)``, each a list over tasks of [B, ...] tensors (heatmaps are channel
        slices of one [B, total_classes, H, W] buffer)."""
        import ctypes
        from ... import _lib
        cfg = self.train_cfg
        lib = _lib.load()
        boxes = [getattr(b, 'tensor', b) for b in gt_bboxes_3d]
        dev = next((b.device for b in boxes if b.is_cuda), None) or next(self.parameters()).device
        assert dev.type == 'cuda', "target assignment runs on the GPU (no CPU fallback)"
        B = len(boxes)
        n_max = max([int(b.shape[0]) for b in boxes] + [1])
        bx = torch.zeros(B, n_max, 9, dtype=torch.float32, device=dev)
        lb = torch.full((B, n_max), -1, dtype=torch.int32, device=dev)
        for i, (b, l) in enumerate(zip(boxes, gt_labels_3d)):
            n = int(b.shape[0])
            if n:
                bx[i, :n] = b.to(dev, torch.float32).reshape(n, -1)[:, :9]
                lb[i, :n] = l.to(dev, torch.int32).reshape(n)
        osf = int(cfg['out_size_factor'])
        W, H = int(cfg['grid_size'][0]) // osf, int(cfg['grid_size'][1]) // osf
        max_objs = int(cfg['max_objs']) * int(cfg['dense_reg'])
        T, total = len(self.num_classes), sum(self.num_classes)
        heat = torch.empty(B, total, H, W, dtype=torch.float32, device=dev)
        anno = torch.empty(T, B, max_objs, 10, dtype=torch.float32, device=dev)
        ind = torch.empty(T, B, max_objs, dtype=torch.int64, device=dev)
        mask = torch.empty(T, B, max_objs, dtype=torch.uint8, device=dev)
        cpt = (ctypes.c_int32 * T)(*self.num_classes)
        with torch.cuda.device(dev), hip_ops.prof("centerhead_targets"):
            rc = lib.sgv3d_centerhead_targets(
                B, n_max, bx.data_ptr(), lb.data_ptr(), T, cpt, max_objs, H, W,
                float(cfg['point_cloud_range'][0]), float(cfg['point_cloud_range'][1]), float(cfg['voxel_size'][0]),
                float(cfg['voxel_size'][1]), float(osf), float(cfg['gaussian_overlap']), int(cfg['min_radius']),
                1 if self.norm_bbox else 0, heat.data_ptr(), anno.data_ptr(), ind.data_ptr(), mask.data_ptr(),
                _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_centerhead_targets")
        heatmaps, c0 = [], 0
        for nc in self.num_classes:
            heatmaps.append(heat[:, c0:c0 + nc])
            c0 += nc
        return heatmaps, [anno[t] for t in range(T)], [ind[t] for t in range(T)], [mask[t] for t in range(T)]

    def loss(self, targets, preds_dicts, **kwargs):
        """Detection loss of bev_height_head.py:255-311 (Gaussian focal loss on the clipped sigmoid heatmap + weighted
        L1 on the gathered box code, summed over tasks) as HIP kernels.  The two averaging factors per task stay on
        the device; with an initialised process group they are averaged over the ranks by ONE all-reduce for all
        tasks (the reference's ``reduce_mean`` per factor, with an ``.item()`` each).

        Returns a scalar tensor.  Gradients with respect to the prediction maps are computed in the same launches;
        they flow to ``preds_dicts`` tensors that require grad, and are kept in ``self.last_pred_grads`` (list over
        tasks of dicts) for the hand-written backward path.  Unlike the reference this does not overwrite
        ``preds_dict[0]['heatmap']`` with its sigmoid nor add an ``'anno_box'`` entry."""
        return _CenterHeadLoss.run(self, targets, preds_dicts)

    def decode_digest(self):
        """Everything ``decode_device`` bakes into kernel arguments, as one hashable value: a recorded decode (BEVHeight's
        hipGraph) is only valid for the configuration it was captured with."""
        def freeze(v):
            if isinstance(v, dict):
                return tuple(sorted((str(k), freeze(x)) for k, x in v.items()))
            if isinstance(v, (list, tuple)):
                return tuple(freeze(x) for x in v)
            if torch.is_tensor(v):
                return tuple(v.flatten().tolist())
            return v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)
        return (freeze(self.bbox_coder_cfg), freeze(self.test_cfg), bool(self.norm_bbox), tuple(int(v) for v in self.num_classes))

    def decode_device(self, preds_dicts):
        """Device half of ``get_bboxes``: top-K / box assembly / circle NMS of every task (three launches) and the merge of
        the tasks (one launch).  Only enqueues kernels on the current stream -- graph-capturable; ``BEVHeight``'s
        per-signature hipGraph runs it right behind the head, so that a harness calling ``get_bboxes`` on the forward's own
        output finds the boxes already decoded (``models/bev_height.py``).

        Returns one packed buffer ``[boxes f32 [B, T*K, 9] | scores f32 [B, T*K] | labels i32 [B, T*K] | counts i32 [B]]`` as a
        uint8 tensor; ``decode_views`` carves it.  Rows ``[:counts[b]]`` of sample ``b`` are its detections."""
        import ctypes
        from ... import _lib
        coder, tcfg = self.bbox_coder_cfg, self.test_cfg
        assert tcfg.get('nms_type', 'circle') == 'circle', "only nms_type='circle' (every shipped config) is built"
        lib = _lib.load()
        K = int(coder['max_num'])
        rng = coder.get('post_center_range')
        rng_c = (ctypes.c_float * 6)(*[float(v) for v in rng]) if rng is not None else None
        thr = coder.get('score_threshold')
        T = len(preds_dicts)
        heat0 = preds_dicts[0][0]['heatmap']
        B, dev = int(heat0.shape[0]), heat0.device
        H, W, bs = int(heat0.shape[2]), int(heat0.shape[3]), int(heat0.stride(0))
        ptrs = {k: (ctypes.c_void_p * T)() for k in ('heatmap', 'reg', 'height', 'dim', 'rot', 'vel')}
        cats = (ctypes.c_int32 * T)()
        has_vel = all(p[0].get('vel') is not None for p in preds_dicts)
        for task_id, preds in enumerate(preds_dicts):
            p = preds[0]
            heat = p['heatmap']
            Bt, cat, Ht, Wt = (int(v) for v in heat.shape)
            assert (Bt, Ht, Wt) == (B, H, W) and heat.device == dev and heat.dtype == torch.float32
            for k in ('heatmap', 'reg', 'height', 'dim', 'rot') + (('vel',) if has_vel else ()):
                assert p[k].stride(0) == bs and p[k].stride(1) == H * W and p[k].stride(3) == 1 and p[k].stride(2) == W
                ptrs[k][task_id] = p[k].data_ptr()
            cats[task_id] = cat
        max_cat = max(int(c) for c in cats)
        # one scratch allocation: [T,B,K] candidates of every task (boxes | scores | labels | valid | keep) + the kernels' workspace
        nws = lib.sgv3d_centerpoint_decode_tasks_workspace_bytes(B, T, max_cat, K)
        n = T * B * K
        scratch = torch.empty(n * (36 + 4 + 4 + 1 + 1) + 64 + nws, dtype=torch.uint8, device=dev)
        base = scratch.data_ptr()
        boxes, scores, labels, valid = base, base + n * 36, base + n * 40, base + n * 44
        keep = valid + n
        ws = (keep + n + 63) // 64 * 64
        nms = (ctypes.c_float * T)(*[float(tcfg['min_radius'][t]) for t in range(T)])
        with torch.cuda.device(dev), hip_ops.prof("centerpoint_decode"):
            rc = lib.sgv3d_centerpoint_decode_tasks(
                B, T, cats, H, W, K, ptrs['heatmap'], ptrs['reg'], ptrs['height'], ptrs['dim'], ptrs['rot'],
                ptrs['vel'] if has_vel else None, bs, float(coder['out_size_factor']), float(coder['voxel_size'][0]),
                float(coder['voxel_size'][1]), float(coder['pc_range'][0]), float(coder['pc_range'][1]),
                float(thr) if thr is not None else float('-inf'), rng_c, 1 if self.norm_bbox else 0, nms,
                int(tcfg['post_max_size']), ws, nws, boxes, scores, labels, valid, keep, _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_centerpoint_decode_tasks")
        # merge tasks (CenterHead.get_bboxes tail): per sample the kept boxes task after task, label offsets, z -= h/2
        TK = T * K
        packed = torch.empty(B * TK * 44 + B * 4, dtype=torch.uint8, device=dev)
        pb = packed.data_ptr()
        ncls = (ctypes.c_int32 * T)(*[int(v) for v in self.num_classes])
        with torch.cuda.device(dev), hip_ops.prof("centerpoint_merge_tasks"):
            rc = lib.sgv3d_centerpoint_merge_tasks(B, T, K, boxes, scores, labels, keep, ncls, pb, pb + B * TK * 36, pb + B * TK * 40,
                                                   pb + B * TK * 44, _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_centerpoint_merge_tasks")
        return packed

    def decode_views(self, packed, B):
        """(boxes [B, T*K, 9] f32, scores [B, T*K] f32, labels [B, T*K] i32, counts [B] i32) views of ``decode_device``'s buffer."""
        TK = (packed.numel() - 4 * B) // (44 * B)
        f = packed[:B * TK * 40].view(torch.float32)
        i = packed[B * TK * 40:].view(torch.int32)
        return f[:B * TK * 9].view(B, TK, 9), f[B * TK * 9:].view(B, TK), i[:B * TK].view(B, TK), i[B * TK:]

    def get_bboxes(self, preds_dicts, img_metas=None, img=None, rescale=False, decoded=None):
        """mmdet3d ``CenterHead.get_bboxes`` (reached via models/bev_height.py:116-126) on the device:
        sigmoid + top-K + box assembly + circle NMS of all tasks and their merge run as HIP kernels (the reference does the
        NMS on the CPU through numba, with a device->host copy per task); the per-sample detection counts are the single
        device->host read of the call.  ``decoded``: a ``decode_device`` buffer computed earlier for these very maps.

        Returns ``[[bboxes, scores, labels], ...]`` per sample; ``bboxes`` is
        ``img_metas[i]['box_type_3d'](tensor, code_size)`` when the harness passes mmdet3d's box class,
        else a ``Boxes3D`` stand-in with the same ``.tensor`` attribute (exps/...:254)."""
        B = int(preds_dicts[0][0]['heatmap'].shape[0])
        packed = decoded if decoded is not None else self.decode_device(preds_dicts)
        out_boxes, out_scores, out_labels, counts = self.decode_views(packed, B)
        n_kept = counts.tolist()
        ret_list = []
        code_size = int(self.bbox_coder_cfg.get('code_size', 9))
        for i in range(B):
            bboxes = out_boxes[i, :n_kept[i]]
            box_type = img_metas[i].get('box_type_3d') if img_metas is not None and i < len(img_metas) else None
            bboxes = box_type(bboxes, code_size) if callable(box_type) else Boxes3D(bboxes, code_size)
            ret_list.append([bboxes, out_scores[i, :n_kept[i]], out_labels[i, :n_kept[i]]])
        return ret_list


_MAP_KEYS = ('heatmap', 'reg', 'height', 'dim', 'rot', 'vel')


class _CenterHeadLoss(torch.autograd.Function):
    """Loss value + gradients of all tasks; autograd only routes the precomputed gradients."""

    @staticmethod
    def run(head, targets, preds_dicts):
        maps = [preds[0][k] for preds in preds_dicts for k in _MAP_KEYS]
        return _CenterHeadLoss.apply(head, targets, *maps)

    @staticmethod
    def forward(ctx, head, targets, *maps):
        import ctypes
        import torch.distributed as dist
        from ... import _lib
        lib = _lib.load()
        heatmaps, anno_boxes, inds, masks = targets
        T = len(heatmaps)
        dev = maps[0].device
        assert dev.type == 'cuda', "the loss runs on the GPU (no CPU fallback)"
        cw = (ctypes.c_float * 10)(*[float(v) for v in head.train_cfg['code_weights']])
        box_w = float(head.loss_bbox_cfg.get('loss_weight', 1.0)) if head.loss_bbox_cfg else 1.0
        cls_w = float(head.loss_cls_cfg.get('loss_weight', 1.0)) if head.loss_cls_cfg else 1.0
        assert cls_w == 1.0, "GaussianFocalLoss loss_weight != 1 is not used by any shipped config"
        stats = torch.empty(T, 2, dtype=torch.float32, device=dev)
        out = torch.empty(T, 2, dtype=torch.float32, device=dev)
        B = int(maps[0].shape[0])
        nws = lib.sgv3d_centerhead_loss_workspace_bytes(B)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        stream = _lib.stream_handle(dev)
        per_task = []
        with torch.cuda.device(dev):
            for t in range(T):
                p = [m if (m.dtype == torch.float32 and m.stride(3) == 1 and m.stride(2) == m.shape[3]
                           and m.stride(1) == m.shape[2] * m.shape[3]) else m.float().contiguous()
                     for m in maps[t * 6:t * 6 + 6]]
                bs = int(p[0].stride(0))
                if any(int(m.stride(0)) != bs for m in p):        # separate tensors: give them a common batch stride
                    buf = torch.cat(p, 1)
                    c0, q = 0, []
                    for m in p:
                        q.append(buf[:, c0:c0 + m.shape[1]])
                        c0 += m.shape[1]
                    p, bs = q, int(buf.stride(0))
                tgt = heatmaps[t]
                assert tgt.dtype == torch.float32 and tgt.stride(3) == 1 and tgt.stride(1) == tgt.shape[2] * tgt.shape[3]
                _, cat, H, W = (int(v) for v in tgt.shape)
                mo = int(inds[t].shape[1])
                m8, i64, an = masks[t].contiguous(), inds[t].contiguous(), anno_boxes[t].contiguous()
                assert m8.dtype == torch.uint8 and i64.dtype == torch.int64 and an.dtype == torch.float32
                with hip_ops.prof("centerhead_loss_stats"):
                    rc = lib.sgv3d_centerhead_loss_stats(B, cat, H, W, mo, tgt.data_ptr(), int(tgt.stride(0)), m8.data_ptr(),
                                                         stats[t].data_ptr(), ws.data_ptr(), nws, stream)
                _lib.check(rc, "sgv3d_centerhead_loss_stats")
                per_task.append((p, bs, tgt, cat, H, W, mo, m8, i64, an))
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(stats)                              # reduce_mean of every factor in one collective
                stats /= dist.get_world_size()
            grads = []
            for t, (p, bs, tgt, cat, H, W, mo, m8, i64, an) in enumerate(per_task):
                gbuf = torch.empty(B, cat + 10, H, W, dtype=torch.float32, device=dev)   # heatmap, reg, height, dim, rot, vel
                g, c0 = [], 0
                for m in p:
                    g.append(gbuf[:, c0:c0 + m.shape[1]])
                    c0 += int(m.shape[1])
                with hip_ops.prof("centerhead_loss"):
                    rc = lib.sgv3d_centerhead_loss(
                        B, cat, H, W, mo, *[m.data_ptr() for m in p], bs, tgt.data_ptr(), int(tgt.stride(0)), an.data_ptr(),
                        i64.data_ptr(), m8.data_ptr(), stats[t].data_ptr(), cw, box_w, 1.0, *[m.data_ptr() for m in g],
                        int(gbuf.stride(0)), out[t].data_ptr(), ws.data_ptr(), nws, stream)
                _lib.check(rc, "sgv3d_centerhead_loss")
                grads.append(g)
        head.last_pred_grads = [dict(zip(_MAP_KEYS, g)) for g in grads]
        head.last_loss_parts = out
        ctx.grads = grads
        ctx.shapes = [m.shape for m in maps]
        return out.sum()

    @staticmethod
    def backward(ctx, grad_out):
        res = [None, None]
        for g in ctx.grads:
            for m in g:
                res.append(m * grad_out)
        return tuple(res)


class Boxes3D:
    """Minimal stand-in for mmdet3d ``LiDARInstance3DBoxes`` (absent in this image): the harness only
    reads ``.tensor`` (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:254)."""

    def __init__(self, tensor, box_dim=9):
        self.tensor = tensor
        self.box_dim = box_dim

    def __len__(self):
        return self.tensor.shape[0]
