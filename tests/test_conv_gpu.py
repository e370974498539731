"""GPU: MFMA implicit-GEMM convolution family and the small layers against plain PyTorch fp32 on CPU
(F.conv2d / F.conv_transpose2d / F.max_pool2d ...: the fp32 reference of a floating-point kernel).
Tolerance: 1e-3 (BASELINE north_star, fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-3, atol=1e-3)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _conv_case(B, cin, H, W, cout, k, stride, pad, dil, seed, **kw):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    return x, w


CASES = [
    # (B, cin, H, W, cout, k, stride, pad, dil)
    (1, 64, 20, 24, 64, 1, 1, 0, 1),
    (2, 64, 17, 19, 128, 3, 1, 1, 1),      # ragged M, 3x3
    (1, 128, 16, 16, 256, 3, 2, 1, 1),     # stride 2
    (1, 256, 14, 18, 96, 3, 1, 6, 6),      # dilated (ASPP), cout not a tile multiple
    (1, 80, 32, 32, 160, 7, 2, 3, 1),      # BEV trunk stem, cin % 32 != 0
    (1, 4, 40, 56, 64, 7, 2, 3, 1),        # image stem with cin padded 3 -> 4
    (1, 256, 12, 12, 128, 4, 4, 0, 1),     # SECONDFPN patchify conv (k == stride)
    (1, 512, 9, 11, 18, 3, 1, 1, 1),       # conv_offset: tiny cout
    (1, 2560, 6, 8, 512, 1, 1, 0, 1),      # ASPP conv1: long K
    (3, 32, 5, 7, 40, 1, 2, 0, 1),         # 1x1 stride 2 (downsample)
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_conv_matches_torch(hip, case, tile):
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout, k, stride, pad, dil = case
    x, w = _conv_case(*case, seed=sum(case))
    ref = F.conv2d(x, w, None, stride, pad, dil)
    conv = PackedConv(w.to(DEV), stride=stride, pad=pad, dil=dil, tile=tile)
    y = conv(nhwc(x).to(DEV))
    torch.cuda.synchronize()
    torch.testing.assert_close(nchw(y.cpu()), ref, **TOL)


def test_conv_exact_on_integers(hip):
    """fp32 MFMA is an exact fp32 FMA chain: small-integer data must come out bit-exact."""
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-3, 4, (1, 64, 13, 15), generator=g).float()
    w = torch.randint(-2, 3, (96, 64, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, 1, 1)
    y = PackedConv(w.to(DEV), pad=1)(nhwc(x).to(DEV))
    assert torch.equal(nchw(y.cpu()), ref)


def test_conv_epilogue_bn_residual_relu_gate_offsets(hip):
    from sgv3d_amd.hip_ops import PackedConv, fold_bn
    g = torch.Generator().manual_seed(7)
    B, cin, H, W, cout = 2, 64, 10, 12, 96
    xfull = torch.randn(B, cin + 32, H, W, generator=g)              # read a channel slice
    w = torch.randn(cout, cin, 3, 3, generator=g) / 24
    bias = torch.randn(cout, generator=g)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g))
        bn.running_mean.copy_(torch.randn(cout, generator=g))
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    bn.eval()
    res = torch.randn(B, cout, H, W, generator=g)
    gate = torch.rand(B, cout, generator=g)
    with torch.no_grad():
        ref = F.relu(bn(F.conv2d(xfull[:, 16:16 + cin], w, bias, 1, 1)) + res) * gate[:, :, None, None]
    scale, shift = fold_bn(bn, bias)
    conv = PackedConv(w.to(DEV), pad=1, scale=scale, shift=shift, relu=True)
    out = torch.full((B, H, W, cout + 40), 7.0, device=DEV)           # write into a concat slice
    conv(nhwc(xfull).to(DEV), out, x_coff=16, y_coff=8, residual=nhwc(res).to(DEV), gate=gate.to(DEV))
    torch.cuda.synchronize()
    o = out.cpu()
    torch.testing.assert_close(nchw(o[..., 8:8 + cout]), ref, **TOL)
    assert (o[..., :8] == 7).all() and (o[..., 8 + cout:] == 7).all()


@pytest.mark.parametrize("ks,cin,cout", [(1, 128, 1024 // 8), (2, 160, 64), (4, 320, 64), (8, 640, 64)])
def test_deconv_kernel_eq_stride(hip, ks, cin, cout):
    """SECONDFPN deblocks: ConvTranspose2d(k = stride) + BN + ReLU, written into a concat slice."""
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(ks)
    x = torch.randn(2, cin, 5, 6, generator=g)
    w = torch.randn(cin, cout, ks, ks, generator=g) / cin ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, None, stride=ks) * scale[None, :, None, None] + shift[None, :, None, None])
    conv = PackedConv(w.to(DEV), stride=ks, transposed=True, scale=scale, shift=shift, relu=True)
    out = torch.zeros(2, 5 * ks, 6 * ks, 256, device=DEV)
    conv(nhwc(x).to(DEV), out, y_coff=64)
    torch.cuda.synchronize()
    torch.testing.assert_close(nchw(out.cpu()[..., 64:64 + cout]), ref, **TOL)


def test_group_planes_mode(hip):
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 9, 10, generator=g)
    w = torch.randn(192, 64, 3, 3, generator=g) / 24
    ref = F.relu(F.conv2d(x, w, None, 1, 1))
    y = PackedConv(w.to(DEV), pad=1, relu=True)(nhwc(x).to(DEV), group_planes=64)     # [3, B, H, W, 64]
    torch.cuda.synchronize()
    assert tuple(y.shape) == (3, 2, 9, 10, 64)
    torch.testing.assert_close(y.cpu().permute(1, 0, 4, 2, 3).reshape(2, 192, 9, 10), ref, **TOL)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_group_planes_fast_epilogue(hip, tile, mode):
    """Whole m-tiles take the row-linear store path of the GROUP_PLANES layout (a wave's columns inside one 64-wide group),
    the ragged tail the general one; 192 channels = 3 groups, so a 128-wide n-tile also carries columns past N."""
    from sgv3d_amd import hip_ops
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(40 + tile)
    x = torch.randn(2, 64, 17, 24, generator=g)                      # M = 816: six / twelve whole tiles + a tail
    w = torch.randn(192, 64, 3, 3, generator=g) / 24
    scale, shift = torch.rand(192, generator=g) + 0.5, torch.randn(192, generator=g)
    old = hip_ops.MFMA_BF16
    hip_ops.MFMA_BF16 = mode == "bf16"
    try:
        y = PackedConv(w.to(DEV), pad=1, relu=True, scale=scale.to(DEV), shift=shift.to(DEV))(nhwc(x).to(DEV), group_planes=64, tile=tile, split_k=1)
    finally:
        hip_ops.MFMA_BF16 = old
    xr, wr = (x.bfloat16().float(), w.bfloat16().float()) if mode == "bf16" else (x, w)
    ref = F.relu(F.conv2d(xr, wr, None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None])
    assert tuple(y.shape) == (3, 2, 17, 24, 64)
    torch.testing.assert_close(y.cpu().permute(1, 0, 4, 2, 3).reshape(2, 192, 17, 24), ref, rtol=1e-4, atol=1e-4)


def test_nchw_out_mode(hip):
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 9, 10, generator=g)
    w = torch.randn(70, 64, 3, 3, generator=g) / 24
    b = torch.randn(70, generator=g)
    ref = F.conv2d(x, w, b, 1, 1)
    y = PackedConv(w.to(DEV), pad=1, shift=b)(nhwc(x).to(DEV), nchw_out=True)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.cpu(), ref, **TOL)


def test_layer_sized_conv_heightnet(hip):
    """The hottest shape of the model: 512->512 3x3 @54x96 (HeightNet, 12.2 GMAC)."""
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 512, 54, 96, generator=g)
    w = torch.randn(512, 512, 3, 3, generator=g) / (512 * 9) ** 0.5
    ref = F.conv2d(x, w, None, 1, 1)
    y = PackedConv(w.to(DEV), pad=1)(nhwc(x).to(DEV))
    torch.cuda.synchronize()
    torch.testing.assert_close(nchw(y.cpu()), ref, **TOL)


def test_channel_padding(hip):
    """cin / cout that are not multiples of 4 (SGV3D's 87 / 174 channels): zero-padded weights."""
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 87, 9, 11, generator=g)
    w = torch.randn(174, 87, 3, 3, generator=g) / 28
    ref = F.relu(F.conv2d(x, w, None, 1, 1))
    xp = torch.cat([nhwc(x), torch.full((1, 9, 11, 1), 3.0)], -1)        # 88-channel buffer, junk in the pad channel
    conv = PackedConv(w.to(DEV), pad=1, relu=True, pad_out=True)
    assert conv.cin == 88 and conv.cout == 176 and conv.cout_real == 174
    y = conv(xp.to(DEV)).cpu()
    torch.testing.assert_close(nchw(y[..., :174]), ref, **TOL)
    assert (y[..., 174:] == 0).all()


def test_conv_rejects_bad_args(hip):
    from sgv3d_amd.hip_ops import PackedConv
    from sgv3d_amd._lib import SGV3DError
    with pytest.raises(SGV3DError):
        PackedConv(torch.randn(8, 6, 1, 1).to(DEV))(torch.zeros(1, 4, 4, 6, device=DEV))   # buffer narrower than cin_pad
    conv = PackedConv(torch.randn(8, 8, 3, 3).to(DEV), pad=1)
    with pytest.raises(SGV3DError):
        conv(torch.zeros(1, 4, 4, 8, device=DEV), torch.zeros(1, 5, 4, 8, device=DEV))   # wrong out size


# ------------------------------------------------------------------------------------------ small layers
def test_maxpool(hip):
    from sgv3d_amd.hip_ops import maxpool3x3s2
    x = torch.randn(2, 64, 11, 14)
    y = maxpool3x3s2(nhwc(x).to(DEV))
    assert torch.equal(nchw(y.cpu()), F.max_pool2d(x, 3, 2, 1))


def test_layout_kernels(hip):
    from sgv3d_amd.hip_ops import nchw_to_nhwc, nhwc_to_nchw
    x = torch.randn(2, 3, 37, 41)
    y = nchw_to_nhwc(x.to(DEV), c_pad=4).cpu()
    assert torch.equal(y[..., :3], nhwc(x)) and (y[..., 3] == 0).all()
    z = torch.randn(2, 9, 13, 70)
    assert torch.equal(nhwc_to_nchw(z.to(DEV), channels=33, coff=5).cpu(), nchw(z[..., 5:38]))


def test_avgpool_dense_broadcast(hip):
    from sgv3d_amd.hip_ops import global_avgpool, dense, broadcast_channels, ACT_RELU, ACT_SIGMOID
    x = torch.randn(2, 512, 9, 7)
    p = global_avgpool(nhwc(x).to(DEV))
    torch.testing.assert_close(p.cpu(), x.mean((2, 3)), rtol=1e-5, atol=1e-5)
    w, b, s = torch.randn(100, 512) / 22, torch.randn(100), torch.rand(100) + 0.5
    torch.testing.assert_close(dense(p, w.to(DEV), s.to(DEV), b.to(DEV), ACT_RELU).cpu(),
                               F.relu((p.cpu() @ w.T) * s + b), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dense(p, w.to(DEV), None, b.to(DEV), ACT_SIGMOID).cpu(),
                               torch.sigmoid(p.cpu() @ w.T + b), rtol=1e-4, atol=1e-4)
    v = torch.randn(2, 48)
    out = torch.zeros(2, 5, 6, 100, device=DEV)
    broadcast_channels(v.to(DEV), out, y_coff=20)
    o = out.cpu()
    assert torch.equal(o[..., 20:68], v[:, None, None, :].expand(2, 5, 6, 48)) and (o[..., :20] == 0).all()


def test_bsm_kernels(hip):
    from sgv3d_amd.hip_ops import upsample_bilinear2x, add_mul_sigmoid, bsm_compose
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 64, 7, 9, generator=g)
    up = upsample_bilinear2x(nhwc(x).to(DEV))
    torch.testing.assert_close(nchw(up.cpu()), F.interpolate(x, scale_factor=2, mode='bilinear'), rtol=1e-5, atol=1e-5)
    a, b, c = (torch.randn(2, 5, 6, 32, generator=g) for _ in range(3))
    torch.testing.assert_close(add_mul_sigmoid(a.to(DEV), b.to(DEV), c.to(DEV)).cpu(), a + b * torch.sigmoid(c),
                               rtol=1e-5, atol=1e-5)
    D, ctx, sem, ld = 12, 80, 7, 12 + 88
    buf = torch.randn(2, 4, 5, ld, generator=g)
    logits = torch.randn(2, 4, 5, sem, generator=g) * 3
    out = bsm_compose(buf.clone().to(DEV), logits.to(DEV), D, ctx, sem, 0.45).cpu()
    p = logits.softmax(-1)
    keep = (1 - (p[..., :1] > 0.45).int()).float()
    assert torch.equal(out[..., :D], buf[..., :D])
    torch.testing.assert_close(out[..., D:D + ctx], buf[..., D:D + ctx] * keep, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[..., D + ctx:D + ctx + sem], p * keep, rtol=1e-5, atol=1e-6)
    assert (out[..., D + ctx + sem:] == 0).all() and (keep == 0).any() and (keep == 1).any()


def _deform_conv_ref(x, offset, weight, groups):
    """DCNv1 (mmcv DeformConv2dPack semantics: 3x3, pad 1, zero padding bilinear) in plain torch fp64."""
    B, C, H, W = x.shape
    x = x.double()
    cols = []
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    for t in range(9):
        ky, kx = t // 3, t % 3
        hf = ys[None] - 1 + ky + offset[:, 2 * t].double()
        wf = xs[None] - 1 + kx + offset[:, 2 * t + 1].double()
        valid = (hf > -1) & (wf > -1) & (hf < H) & (wf < W)
        h0, w0 = torch.floor(hf), torch.floor(wf)
        lh, lw = hf - h0, wf - w0
        val = torch.zeros(B, C, H, W, dtype=torch.float64)
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = (h0 + dh).long(), (w0 + dw).long()
            ok = valid & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1))[:, None].expand(B, C, H, W).reshape(B, C, -1)
            g = torch.gather(x.reshape(B, C, -1), 2, idx).reshape(B, C, H, W)
            val = val + g * (wt * ok)[:, None]
        cols.append(val)
    col = torch.stack(cols, 2)                                   # [B, C, 9, H, W]
    cout = weight.shape[0]
    cpg, opg = C // groups, cout // groups
    out = torch.zeros(B, cout, H, W, dtype=torch.float64)
    for g_ in range(groups):
        wg = weight[g_ * opg:(g_ + 1) * opg].double().reshape(opg, cpg * 9)
        cg = col[:, g_ * cpg:(g_ + 1) * cpg].reshape(B, cpg * 9, H * W)
        out[:, g_ * opg:(g_ + 1) * opg] = (wg @ cg).reshape(B, opg, H, W)
    return out.float()


def test_deform_conv(hip):
    """deform_im2col + grouped GEMM == DCNv1; zero offsets == plain grouped conv (known answer)."""
    from sgv3d_amd.hip_ops import PackedConv, deform_im2col3x3
    g = torch.Generator().manual_seed(5)
    B, C, H, W, groups = 2, 64, 9, 11, 4
    x = torch.randn(B, C, H, W, generator=g)
    weight = torch.randn(C, C // groups, 3, 3, generator=g) / 12
    for offs in (torch.zeros(B, 18, H, W), torch.randn(B, 18, H, W, generator=g) * 2.5):
        col = deform_im2col3x3(nhwc(x).to(DEV), nhwc(offs).to(DEV), groups)
        out = torch.empty(B, H, W, C, device=DEV)
        cpg = C // groups
        for gi in range(groups):
            wg = weight[gi * cpg:(gi + 1) * cpg].permute(0, 2, 3, 1).reshape(cpg, 9 * cpg, 1, 1)   # [o, (tap, c)]
            PackedConv(wg.contiguous().to(DEV))(col, out, x_coff=gi * 9 * cpg, y_coff=gi * cpg)
        torch.cuda.synchronize()
        ref = _deform_conv_ref(x, offs, weight, groups)
        torch.testing.assert_close(nchw(out.cpu()), ref, **TOL)
        if offs.abs().sum() == 0:
            torch.testing.assert_close(ref, F.conv2d(x, weight, None, 1, 1, 1, groups), **TOL)


@pytest.mark.parametrize("shape", [(2, 128, 9, 11, 4, 128), (1, 512, 54, 96, 4, 512), (1, 256, 13, 7, 2, 96), (3, 64, 5, 5, 2, 264)])
def test_deform_conv_fused(hip, shape):
    """sgv3d_deform_conv3x3_forward (csrc/dcn_fused.hip): the deformable 3x3 of lss_fpn.py:190-198 as one implicit GEMM with the
    bilinear samples formed on the way into LDS -- against the float64 DCNv1 restatement and against the im2col + per-group
    GEMM form it replaces (same samples, another k order: f32 rounding only).  Offsets up to +-2.5 pixels plus a few that throw
    the sample far outside the image; zero offsets = the plain grouped convolution; pixel counts that are not multiples of the
    64-row tile; an output written at a channel offset of a wider buffer; out-of-group channel counts above one 64-column tile."""
    from sgv3d_amd.hip_ops import PackedConv, deform_conv3x3, deform_conv3x3_eligible, deform_im2col3x3
    B, C, H, W, groups, cout = shape
    g = torch.Generator().manual_seed(50 + C)
    x = torch.randn(B, C, H, W, generator=g)
    weight = torch.randn(cout, C // groups, 3, 3, generator=g) / (9 * C // groups) ** 0.5
    cpg, opg = C // groups, cout // groups
    convs = [PackedConv(weight[gi * opg:(gi + 1) * opg].permute(0, 2, 3, 1).reshape(opg, 9 * cpg, 1, 1).contiguous().to(DEV))
             for gi in range(groups)]
    xd = nhwc(x).to(DEV)
    assert deform_conv3x3_eligible(xd, convs)
    big = torch.randn(B, 18, H, W, generator=g) * 2.5
    big[:, :, 0, 0] = 40.0                                      # far outside: zeros
    big[:, :, H - 1, W - 1] = -40.0
    for offs in (torch.zeros(B, 18, H, W), big):
        od = nhwc(offs).to(DEV)
        out = torch.full((B, H, W, cout + 12), 7.0, device=DEV)
        deform_conv3x3(xd, od, convs, out=out, y_coff=8)
        torch.cuda.synchronize()
        assert float(out[..., :8].min()) == 7.0 and float(out[..., cout + 8:].min()) == 7.0          # neighbours untouched
        got = out[..., 8:cout + 8]
        ref = _deform_conv_ref(x, offs, weight, groups)
        torch.testing.assert_close(nchw(got.cpu()), ref, **TOL)
        if offs.abs().sum() == 0:
            torch.testing.assert_close(ref, F.conv2d(x, weight, None, 1, 1, 1, groups), **TOL)
        col = deform_im2col3x3(xd, od, groups)
        old = torch.empty(B, H, W, cout, device=DEV)
        for gi, conv in enumerate(convs):
            conv(col, old, x_coff=gi * 9 * cpg, y_coff=gi * opg)
        scale = float(ref.abs().max())
        assert float((old - got).abs().max()) <= 2e-5 * max(1.0, scale)
        again = torch.full_like(out, 7.0)
        deform_conv3x3(xd, od, convs, out=again, y_coff=8)
        assert torch.equal(out, again)                          # deterministic


def test_head_final_conv(hip):
    from sgv3d_amd.hip_ops import head_final_conv
    g = torch.Generator().manual_seed(9)
    B, H, W, hc = 2, 21, 37, 64
    widths = [2, 1, 3, 2, 2, 1]                                    # reg, height, dim, rot, vel, heatmap
    nb = len(widths)
    hidden = torch.randn(B, nb * hc, H, W, generator=g)
    ws = [torch.randn(c, hc, 3, 3, generator=g) / 24 for c in widths]
    bs = [torch.randn(c, generator=g) for c in widths]
    ref = torch.cat([F.conv2d(hidden[:, i * hc:(i + 1) * hc], ws[i], bs[i], 1, 1) for i in range(nb)], 1)
    wcat = torch.cat([w.permute(0, 2, 3, 1) for w in ws], 0).contiguous()       # [sum_c, 3, 3, hc]
    branch = torch.tensor(sum([[i] * c for i, c in enumerate(widths)], []), dtype=torch.int32)
    planes = hidden.reshape(B, nb, hc, H, W).permute(1, 0, 3, 4, 2).contiguous()      # [nb, B, H, W, hc]
    out = head_final_conv(planes.to(DEV), wcat.to(DEV), torch.cat(bs).to(DEV), branch.to(DEV), nb, hc)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), ref, **TOL)


def test_random_shape_fuzz_all_algorithms():
    """tools/fuzz_conv.py in small: random conv shapes through every algorithm / tile / split-K variant."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_conv.py"), "60", "3"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_autotune_under_load_picks_a_valid_choice(hip):
    """SGV3D_TUNE_STREAMS > 1: candidates are timed as concurrent copies on side streams; whatever wins, the result is
    the same convolution."""
    from sgv3d_amd import hip_ops
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 64, 24, 40, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    old = hip_ops.TUNE_STREAMS
    hip_ops.TUNE_STREAMS = 3
    try:
        conv = PackedConv(w.to(DEV), pad=1, relu=True)
        y = conv(nhwc(x).to(DEV))
        assert len(conv._tile_cache) == 1                       # the measurement ran and its choice is cached
        y2 = conv(nhwc(x).to(DEV))
    finally:
        hip_ops.TUNE_STREAMS = old
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    torch.testing.assert_close(nchw(y.cpu()), F.relu(F.conv2d(x, w, None, 1, 1)), **TOL)


@pytest.mark.parametrize("shape", [(1, 64, 40, 48, 256, 1, 1, 0), (1, 128, 27, 31, 512, 1, 1, 0), (2, 256, 9, 13, 100, 1, 1, 0),
                                   (1, 64, 24, 24, 64, 3, 1, 1), (1, 256, 20, 20, 128, 1, 2, 0)])
@pytest.mark.parametrize("with_res", [False, True])
def test_swapped_operand_epilogue_is_bitwise_the_unswapped_kernel(hip, shape, with_res):
    """64x64 tile, NHWC output, SGV3D_SWAP_EPI=1: the launcher takes the operand-swapped kernel (C^T = W . X^T, 16-byte
    residual loads and stores).  Same products, same k order: bitwise equal to the default kernel, with folded BN,
    residual, ReLU, a concat offset, ragged last m-tile and a channel count that is not a multiple of 32."""
    import os
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
    sc = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    sh = torch.randn(cout, generator=g).to(DEV)
    conv = PackedConv(w, stride=stride, pad=pad, scale=sc, shift=sh, relu=True, tile=4)
    oh, ow = conv.out_hw(H, W)
    res = torch.randn(B, oh, ow, cout, generator=g).to(DEV) if with_res else None
    outs = []
    for flag in (None, "1"):
        if flag is None:
            os.environ.pop("SGV3D_SWAP_EPI", None)
        else:
            os.environ["SGV3D_SWAP_EPI"] = flag
        try:
            out = torch.full((B, oh, ow, cout + 8), -7.0, device=DEV)          # channel-slice write at offset 4
            conv(x, out, y_coff=4, residual=res)
            torch.cuda.synchronize()
            outs.append(out)
        finally:
            os.environ.pop("SGV3D_SWAP_EPI", None)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[1][..., :4].min()) == -7.0 and float(outs[1][..., cout + 4:].max()) == -7.0    # neighbours untouched
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w.cpu(), None, stride, pad) * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None]
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(outs[1][..., 4:cout + 4].permute(0, 3, 1, 2).cpu(), ref.clamp_min(0), **TOL)


@pytest.mark.parametrize("shape", [(1, 160, 40, 48, 256, 4), (1, 128, 27, 31, 512, 4), (2, 256, 9, 13, 100, 3), (1, 1024, 11, 17, 256, 4),
                                   (1, 512, 13, 9, 2048, 3)])
@pytest.mark.parametrize("split", [1, 2])
def test_pointwise_specialisation_is_bitwise_the_generic_kernel(hip, shape, split):
    """1x1 / stride 1 / cin % 32 == 0 layers on the 64x64 (4) and 64x128 (3) tiles take the pointwise instantiation (no vector
    instruction for the A addresses inside the k loop).  Same loads, same order: bitwise equal to the generic kernel
    (SGV3D_NO_PW_KERNEL=1), with folded BN, residual, ReLU, split-K and a ragged last m-tile."""
    import os
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout, tile = shape
    if cin // 32 < split:
        pytest.skip("not enough k-tiles")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
    res = torch.randn(B, H, W, cout, generator=g).to(DEV)
    conv = PackedConv(w, scale=sc, shift=sh, relu=True)
    outs = []
    for flag in ("1", None):
        if flag:
            os.environ["SGV3D_NO_PW_KERNEL"] = flag
        else:
            os.environ.pop("SGV3D_NO_PW_KERNEL", None)
        try:
            outs.append(conv(x, residual=res, tile=tile, split_k=split))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("SGV3D_NO_PW_KERNEL", None)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w.cpu()) * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None] + res.permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(outs[1].permute(0, 3, 1, 2).cpu(), ref.clamp_min(0), **TOL)


def test_tune_db_entry_is_checked_against_the_switches_and_the_layout(hip):
    """A tune-DB entry is a recorded measurement, not an override: with the Winograd kill switches on or with
    SGV3D_NO_AUTOTUNE's fixed rule the entry is not used (a stale one is dropped and the layer re-measured among the
    candidates that are allowed now) -- the call neither raises nor runs the disabled kernel."""
    from sgv3d_amd import hip_ops
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (1, 128, 32, 40), generator=g).float()
    w = torch.randint(-2, 3, (128, 128, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, 1, 1)
    xd = nhwc(x).to(DEV)
    saved = (hip_ops.WINOGRAD, hip_ops.WINO4, hip_ops.AUTOTUNE, dict(hip_ops.TUNE_DB))
    try:
        conv = PackedConv(w.to(DEV), pad=1)
        hip_ops.PROFILE = []
        conv(xd)                                            # measured: leaves its signature in the DB
        sig = next(k for k in hip_ops.TUNE_DB if k not in saved[3] and k.startswith("128x128k3x3"))
        for forced in ((hip_ops.TILE_WINO, 1), (hip_ops.TILE_WINO4, 1)):
            hip_ops.TUNE_DB[sig] = forced
            hip_ops.WINOGRAD, hip_ops.WINO4 = False, False
            c2 = PackedConv(w.to(DEV), pad=1)
            hip_ops.PROFILE = []
            y = c2(xd)
            assert all("wino" not in rec[0] for rec in hip_ops.PROFILE), hip_ops.PROFILE[-1][0]
            assert torch.equal(nchw(y.cpu()), ref)
            assert tuple(hip_ops.TUNE_DB[sig]) != forced                    # dropped and re-measured without Winograd
            hip_ops.WINOGRAD, hip_ops.WINO4 = saved[0], saved[1]
        # the fixed rule ignores the DB altogether
        hip_ops.TUNE_DB[sig] = (hip_ops.TILE_WINO4, 1)
        hip_ops.AUTOTUNE = False
        c3 = PackedConv(w.to(DEV), pad=1)
        hip_ops.PROFILE = []
        y = c3(xd)
        assert hip_ops.PROFILE[-1][0] == "conv_wino" and tuple(hip_ops.TUNE_DB[sig]) == (hip_ops.TILE_WINO4, 1)
        assert torch.equal(nchw(y.cpu()), ref)
        hip_ops.AUTOTUNE = saved[2]
    finally:
        hip_ops.PROFILE = None
        hip_ops.WINOGRAD, hip_ops.WINO4, hip_ops.AUTOTUNE = saved[:3]
        hip_ops.TUNE_DB.clear()
        hip_ops.TUNE_DB.update(saved[3])


@pytest.mark.parametrize("shape", [(1, 64, 54, 96, 256, 1, 1, 0), (2, 256, 27, 33, 64, 1, 1, 0), (1, 128, 20, 31, 512, 1, 1, 0),
                                   (1, 64, 33, 47, 96, 3, 1, 1), (1, 160, 32, 32, 320, 3, 2, 1), (1, 512, 14, 18, 100, 1, 1, 0)])
@pytest.mark.parametrize("split", [1, 2])
def test_five_per_cu_tile_is_bitwise_the_64x64_tile(hip, shape, split):
    """SGV3D_TILE_OCC5 (32 KB of swizzled LDS, one register stage, residual read in the epilogue) sums every output's k in the
    order of the plain 64x64 tile: bitwise the same result, for pointwise and general layers, with residual / BN / ReLU,
    ragged m- and n-tiles, split-K and the m-tile-first walk."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.2).cuda()
    conv = PackedConv(w, stride=stride, pad=pad, scale=sc, shift=sh, relu=True)
    oh, ow = conv.out_hw(H, W)
    res = torch.randn(B, oh, ow, cout, generator=g).cuda()
    if split > conv.k_pad // 32:
        pytest.skip("not enough k-tiles")
    base = conv(x, residual=res, tile=4, split_k=split)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double() + res.double()).clamp_min(0)
    assert float((base.double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    for t in (44, 45):
        if t == 45 and cout <= 64:
            continue
        got = conv(x, residual=res, tile=t, split_k=split)
        assert torch.equal(got, base), (t, float((got - base).abs().max()))
    # plain epilogue (no BN / residual / ReLU): the raw-store path
    plain = PackedConv(w, stride=stride, pad=pad)
    assert torch.equal(plain(x, tile=44, split_k=split), plain(x, tile=4, split_k=split))


# ------------------------------------------------------------------------------------------------ pointwise f32x3 (csrc/conv_pw_x3.hip)
PW_X3_SHAPES = [
    # (B, cin, H, W, cout)
    (1, 64, 40, 48, 256),       # ResNet layer 1 expand (HBM-bound shape)
    (1, 256, 27, 31, 64),       # ragged M (837 pixels: partial last m-tile at every tile height)
    (2, 128, 9, 13, 512),       # batch 2
    (1, 1024, 11, 17, 256),     # long K
    (1, 512, 20, 12, 100),      # cout not a multiple of 32 (rows of the packed weights padded to 128), N tail inside a wave
    (1, 96, 7, 5, 36),          # cin = 3 k-steps, tiny cout
]


@pytest.mark.parametrize("shape", PW_X3_SHAPES)
@pytest.mark.parametrize("tile", [60, 61, 62, 64, 65, 66, 70, 71, 72, 74, 75, 76])
def test_pointwise_x3_matches_float64_like_the_f32_kernel(hip, shape, tile):
    """conv_pw_x3_kernel through PackedConv(tile=...): a 1x1 convolution with folded BN, residual, ReLU, an input channel window and a
    concat offset, against a float64 product -- and against the f32-MFMA implicit GEMM on the same call: the f32x3 form (weights split
    into three bf16 terms by the packer, activations on their way into LDS, six partial products in f32) is as close to float64 as the
    native kernel (bar: 1.5x its error, and 2e-6 of the output scale), on activations whose channels span four decades."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + tile)
    x = torch.randn(B, H, W, cin + 16, generator=g) * torch.pow(10.0, torch.randint(-2, 3, (1, 1, 1, cin + 16), generator=g).float())
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, H, W, cout, generator=g)
    conv = PackedConv(w.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), relu=True)
    assert conv.pw_x3_ok()
    want = x[..., 8:8 + cin].double().reshape(-1, cin) @ w.double().reshape(cout, cin).t() * scale.double() + shift.double()
    want = torch.relu(want + res.double().reshape(-1, cout)).reshape(B, H, W, cout)
    s = float(want.abs().max())
    out = torch.full((B, H, W, cout + 12), -3.0, device=DEV)
    conv(x.to(DEV), out, x_coff=8, y_coff=4, residual=res.to(DEV), tile=tile, split_k=1)
    err = float((out[..., 4:4 + cout].cpu().double() - want).abs().max()) / s
    native = conv(x.to(DEV), x_coff=8, residual=res.to(DEV), tile=4, split_k=1)
    e_native = float((native.cpu().double() - want).abs().max()) / s
    assert err <= max(1.5 * e_native, 2e-7) and err < 2e-6, (err, e_native)
    assert float(out[..., :4].max()) == -3.0 and float(out[..., 4 + cout:].max()) == -3.0          # neighbours untouched
    again = torch.full_like(out, -3.0)
    conv(x.to(DEV), again, x_coff=8, y_coff=4, residual=res.to(DEV), tile=tile, split_k=1)
    assert torch.equal(out, again)                                                                  # deterministic


@pytest.mark.parametrize("case", [(1, 128, 30, 44, 128, 3, 2, 1, 1), (2, 64, 17, 19, 96, 3, 1, 1, 1), (1, 256, 16, 24, 512, 1, 2, 0, 1),
                                  (1, 256, 24, 32, 128, 4, 4, 0, 1), (1, 64, 21, 13, 64, 3, 1, 2, 2), (1, 96, 12, 20, 36, 5, 1, 2, 1),
                                  (1, 512, 9, 11, 128, 2, 2, 0, 1), (3, 32 * 3, 5, 7, 40, 1, 2, 0, 1),
                                  # tap-major weights (cin % 32 != 0, up to 64 taps): the 7x7 stems, and a 3x3 on 20 channels
                                  (1, 80, 32, 32, 160, 7, 2, 3, 1), (2, 4, 40, 56, 64, 7, 2, 3, 1), (1, 20, 13, 17, 48, 3, 1, 1, 1)])
@pytest.mark.parametrize("tile,split", [(60, 1), (61, 1), (62, 1), (65, 1), (71, 1), (76, 1), (60, 2), (64, 3)])
def test_implicit_gemm_x3_strided_and_multi_tap(hip, case, tile, split):
    """The same kernel as an implicit GEMM over taps (TAPS = true): strided 3x3 / 1x1, patchify (k == stride), dilated and 5x5
    layers with ragged maps and padding on every side, with BN / residual / ReLU and split-K, against a float64 convolution and the
    f32-MFMA kernel (as accurate as it: within 2x its error -- the two kernels add the K products in different orders --, < 3e-6 of
    the output scale)."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout, k, stride, pad, dil = case
    x, w = _conv_case(*case, seed=sum(case))
    g = torch.Generator().manual_seed(sum(case) + tile)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    conv = PackedConv(w.to(DEV), stride=stride, pad=pad, dil=dil, scale=scale.to(DEV), shift=shift.to(DEV), relu=True)
    assert conv.pw_x3_ok()
    OH, OW = conv.out_hw(H, W)
    res = torch.randn(B, OH, OW, cout, generator=g)
    if split > 1 and conv.k_pad // 32 < split:
        pytest.skip("not enough k-steps")
    want = F.conv2d(x.double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1) * scale.double() + shift.double()
    want = torch.relu(want + res.double())
    s = float(want.abs().max())
    y = conv(nhwc(x).to(DEV), residual=res.to(DEV), tile=tile, split_k=split)
    native = conv(nhwc(x).to(DEV), residual=res.to(DEV), tile=4, split_k=1)
    err = float((y.cpu().double() - want).abs().max()) / s
    e_native = float((native.cpu().double() - want).abs().max()) / s
    assert err <= max(2.0 * e_native, 3e-7) and err < 3e-6, (err, e_native)


def test_pointwise_x3_keeps_every_partial_product_and_covers_only_what_it_should(hip):
    """Integer data (exact in f32 on both paths, exact under the three-term split): every tile shape returns the f32 kernel's bits --
    a dropped partial product or a swapped plane shows up as a whole number.  And the launch refuses what the kernel does not cover."""
    from sgv3d_amd._lib import SGV3DError
    from sgv3d_amd.hip_ops import PackedConv, PW_X3_TILES
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-200, 201, (1, 19, 23, 160), generator=g).float()
    w = torch.randint(-300, 301, (96, 160, 1, 1), generator=g).float()
    conv = PackedConv(w.to(DEV))
    want = conv(x.to(DEV), tile=4, split_k=1)
    ref = (x.double().reshape(-1, 160) @ w.double().reshape(96, 160).t()).reshape(1, 19, 23, 96)
    assert torch.equal(want.cpu().double(), ref)                          # |sum| < 2^24: exact
    for tile in PW_X3_TILES:
        assert torch.equal(conv(x.to(DEV), tile=tile, split_k=1), want), tile
    for split in (2, 5):                                                  # split-K: partial sums + fixed-order reduce, still exact here
        assert torch.equal(conv(x.to(DEV), tile=61, split_k=split), want), split
    w7 = torch.randn(64, 64, 7, 7)
    with pytest.raises(SGV3DError):
        PackedConv(w7.to(DEV), pad=3)(torch.randn(1, 8, 8, 64, device=DEV), tile=61, split_k=1)     # more than 32 taps
    assert not PackedConv(torch.randn(64, 48, 1, 1).to(DEV)).pw_x3_ok()  # cin % 32 != 0
