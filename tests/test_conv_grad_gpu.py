"""Convolution backward on the MI355X against torch autograd in float64 on the CPU:
weight gradient kernel, data gradient through the forward kernels, and the autograd function."""
import pytest
import torch
import torch.nn.functional as F

from sgv3d_amd import conv_grad

pytestmark = pytest.mark.gpu

# (batch, cin, H, W, cout, k, stride, pad, dil)
SHAPES = [
    (2, 64, 20, 28, 64, 3, 1, 1, 1),        # ResNet layer1 3x3
    (1, 64, 24, 40, 256, 1, 1, 0, 1),       # bottleneck expand
    (2, 128, 21, 27, 128, 3, 2, 1, 1),      # strided 3x3, odd sizes
    (1, 256, 18, 26, 512, 1, 2, 0, 1),      # downsample branch
    (1, 4, 38, 50, 64, 7, 2, 3, 1),         # stem (3 channels padded to 4)
    (1, 128, 20, 24, 128, 3, 1, 6, 6),      # ASPP dilated
    (2, 80, 16, 16, 160, 3, 1, 1, 1),       # BEV trunk widths (not multiples of 64)
    (1, 96, 12, 20, 18, 3, 1, 1, 1),        # DCN offset conv: cout % 4 != 0
    (1, 64, 16, 16, 32, 4, 4, 0, 1),        # neck patchify conv (kernel == stride)
    (2, 32, 19, 22, 48, 2, 2, 0, 1),        # kernel == stride with rows / columns the conv never reads
    (3, 32, 9, 11, 48, 3, 1, 1, 1),         # tiny, several images
    (1, 80, 30, 34, 160, 7, 2, 3, 1),       # BEV trunk stem: 7x7 stride 2 (four-phase data gradient)
    (2, 64, 17, 23, 64, 3, 2, 0, 1),        # stride 2 without padding, odd sizes
    (1, 32, 16, 20, 32, 5, 2, 2, 1),        # 5x5 stride 2
    (1, 32, 15, 19, 64, 3, 3, 1, 1),        # stride 3: zero-insertion path
    (1, 174, 12, 14, 174, 3, 1, 1, 1),      # BSM head width 2 x 87: neither side a multiple of 4
    (1, 174, 12, 14, 348, 3, 2, 1, 1),      # ... and its strided successor
]


def _reference(x, w, dy, stride, pad, dil):
    xr = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wr = w.double().clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad, dil)
    y.backward(dy.double().permute(0, 3, 1, 2))
    return y.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_backward_matches_autograd(shape):
    B, cin, H, W, cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    oh = (H + 2 * p - d * (k - 1) - 1) // s + 1
    ow = (W + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(B, oh, ow, cout, generator=g)
    y_ref, dx_ref, dw_ref = _reference(x, w, dy, s, p, d)
    xg = x.cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = conv_grad.conv2d(xg, wg, None, s, p, d)
    assert float((y.detach().cpu().double() - y_ref).abs().max()) <= 1e-5 * float(y_ref.abs().max())
    y.backward(dy.cuda())
    for name, got, want in (("dx", xg.grad, dx_ref), ("dw", wg.grad, dw_ref)):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()), (name, err, float(want.abs().max()))


@pytest.mark.parametrize("split", [1, 2, 5, 0])
def test_wgrad_split_variants_and_channel_windows(split):
    g = torch.Generator().manual_seed(split)
    B, H, W = 2, 19, 23
    x_full = torch.randn(B, H, W, 96, generator=g)
    dy_full = torch.randn(B, H, W, 80, generator=g)
    x, dy = x_full[..., 16:80], dy_full[..., 8:72]          # 64-channel windows of wider buffers
    w = torch.zeros(64, 64, 3, 3)
    _, _, dw_ref = _reference(x, w, dy, 1, 1, 1)
    dw = conv_grad.conv2d_backward_weight(x_full.cuda(), dy_full.cuda(), 3, 1, 1, 1, cin=64, cout=64, x_coff=16, y_coff=8, split=split)
    assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
    again = conv_grad.conv2d_backward_weight(x_full.cuda(), dy_full.cuda(), 3, 1, 1, 1, cin=64, cout=64, x_coff=16, y_coff=8, split=split)
    assert torch.equal(dw, again)                            # fixed-order reduction


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, cout, pad
    (2, 64, 20, 28, 64, 1),          # one tile, rows shorter than a 32-pixel segment
    (1, 128, 9, 70, 192, 1),         # several tiles, rows of two full segments + a 6-pixel tail
    (3, 80, 16, 33, 160, 1),         # channel counts that are not multiples of 64, a 1-pixel tail segment
    (1, 64, 12, 40, 64, 0),          # no padding: the output is smaller than the input
    (2, 32, 7, 96, 36, 2),           # padding 2: two border rows / columns of zeros
    (1, 512, 6, 8, 512, 1),          # many tiles on a tiny map
])
@pytest.mark.parametrize("split", [0, 1, 3])
def test_wgrad_all_taps_kernel(shape, split):
    """Tile id 5 (conv_wgrad3x3_kernel: the nine taps of a 64 x 64 tile in one workgroup, stages = row segments) against
    float64 autograd; bitwise repeatable; exact on small integers."""
    B, cin, H, W, cout, pad = shape
    g = torch.Generator().manual_seed(sum(shape) + split)
    x = torch.randn(B, H, W, cin + 8, generator=g)
    oh, ow = H + 2 * pad - 2, W + 2 * pad - 2
    dy = torch.randn(B, oh, ow, cout + 4, generator=g)
    _, _, dw_ref = _reference(x[..., 4:4 + cin], torch.zeros(cout, cin, 3, 3), dy[..., 4:], 1, pad, 1)
    dw = conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, pad, 1, cin=cin, cout=cout, x_coff=4, y_coff=4, split=split, tile=5)
    assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
    again = conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, pad, 1, cin=cin, cout=cout, x_coff=4, y_coff=4, split=split, tile=5)
    assert torch.equal(dw, again)
    xi = torch.randint(-3, 4, (B, H, W, cin), generator=g).float()
    dyi = torch.randint(-2, 3, (B, oh, ow, cout), generator=g).float()
    _, _, want = _reference(xi, torch.zeros(cout, cin, 3, 3), dyi, 1, pad, 1)
    assert torch.equal(conv_grad.conv2d_backward_weight(xi.cuda(), dyi.cuda(), 3, 1, pad, 1, split=split, tile=5).cpu().double(), want)


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, cout, k, stride, pad, dil
    (2, 64, 20, 28, 64, 3, 1, 1, 1),          # one 64 x 64 tile per tap
    (1, 256, 18, 26, 512, 1, 2, 0, 1),        # 1x1 stride 2, several 128 x 128 tiles
    (2, 128, 21, 27, 128, 3, 2, 1, 1),        # strided 3x3, odd sizes
    (1, 128, 20, 24, 128, 3, 1, 6, 6),        # dilated: taps far outside the image
    (2, 80, 16, 16, 160, 3, 1, 1, 1),         # channel counts that are not multiples of 64
    (1, 80, 30, 34, 160, 7, 2, 3, 1),         # 7x7 stride 2 (the f32 kernel's flat tap columns do not exist here)
    (3, 32, 9, 11, 48, 3, 1, 1, 1),           # tiny, several images, ragged last stage
    (1, 512, 27, 48, 512, 3, 1, 1, 1),        # HeightNet-like at half size
])
@pytest.mark.parametrize("tile,split", [(1, 1), (1, 3), (4, 1), (4, 5), (0, 0)])
def test_wgrad_bf16_kernel(shape, tile, split):
    """sgv3d_conv2d_backward_weight_bf16 (mixed-precision training): f32 tensors, products of bf16-rounded operands on
    v_mfma_f32_32x32x16_bf16, f32 accumulation.  Against the float64 gradient of the ROUNDED operands: 2e-5 (summation order only);
    against the unrounded gradient: 1e-2 of its scale (two roundings of 2^-9 each, averaged over the pixel sum); exact on small
    integers; bitwise repeatable; channel windows of wider buffers."""
    B, cin, H, W, cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(sum(shape) + 7 * tile + split)
    x = torch.randn(B, H, W, cin + 8, generator=g)
    oh = (H + 2 * p - d * (k - 1) - 1) // s + 1
    ow = (W + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(B, oh, ow, cout + 4, generator=g)
    xs, dys = x[..., 4:4 + cin], dy[..., 4:]
    zero_w = torch.zeros(cout, cin, k, k)
    _, _, dw_exact = _reference(xs, zero_w, dys, s, p, d)
    _, _, dw_rounded = _reference(xs.bfloat16().float(), zero_w, dys.bfloat16().float(), s, p, d)
    run = lambda: conv_grad.conv2d_backward_weight_bf16(x.cuda(), dy.cuda(), k, s, p, d, cin=cin, cout=cout, x_coff=4, y_coff=4,
                                                        split=split, tile=tile)
    dw = run()
    scale = float(dw_exact.abs().max())
    assert float((dw.cpu().double() - dw_rounded).abs().max()) <= 2e-5 * scale
    assert float((dw.cpu().double() - dw_exact).abs().max()) <= 1e-2 * scale
    assert torch.equal(dw, run())
    xi = torch.randint(-3, 4, (B, H, W, cin), generator=g).float()
    dyi = torch.randint(-2, 3, (B, oh, ow, cout), generator=g).float()
    _, _, dwi = _reference(xi, zero_w, dyi, s, p, d)
    got = conv_grad.conv2d_backward_weight_bf16(xi.cuda(), dyi.cuda(), k, s, p, d, split=split, tile=tile)
    assert torch.equal(got.cpu().double(), dwi)


def test_wgrad_all_taps_kernel_rejects_other_layers():
    from sgv3d_amd import _lib
    x, dy = torch.randn(1, 8, 8, 64).cuda(), torch.randn(1, 4, 4, 64).cuda()
    with pytest.raises(_lib.SGV3DError):
        conv_grad.conv2d_backward_weight(x, dy, 3, 2, 1, 1, tile=5, split=1)          # stride 2


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, cout, pad
    (2, 64, 20, 28, 3, 1),           # CenterHead final layer: 64 -> 3
    (1, 64, 9, 300, 1, 1),           # one output channel, rows of three 128-pixel segments (the last one 44 pixels)
    (3, 80, 7, 17, 2, 0),            # cin not a multiple of 64, no padding
    (1, 130, 6, 33, 4, 2),           # three channel tiles, padding 2
])
def test_wgrad_thin_kernel(shape):
    """1..4 output channels (sgv3d_conv2d_backward_weight_thin, taken automatically): against float64 autograd, bitwise
    repeatable, exact on small integers."""
    B, cin, H, W, cout, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, cin + 8, generator=g)
    oh, ow = H + 2 * pad - 2, W + 2 * pad - 2
    dy = torch.randn(B, oh, ow, cout, generator=g)
    _, _, dw_ref = _reference(x[..., 4:4 + cin], torch.zeros(cout, cin, 3, 3), dy, 1, pad, 1)
    dw = conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, pad, 1, cin=cin, cout=cout, x_coff=4)
    assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
    assert torch.equal(dw, conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, pad, 1, cin=cin, cout=cout, x_coff=4))
    assert float((dw - conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, pad, 1, cin=cin, cout=cout, x_coff=4, tile=1, split=4)).abs().max()) \
        <= 2e-5 * float(dw_ref.abs().max())                                   # the MFMA kernel agrees
    xi = torch.randint(-3, 4, (B, H, W, cin), generator=g).float()
    dyi = torch.randint(-2, 3, (B, oh, ow, cout), generator=g).float()
    _, _, want = _reference(xi, torch.zeros(cout, cin, 3, 3), dyi, 1, pad, 1)
    assert torch.equal(conv_grad.conv2d_backward_weight(xi.cuda(), dyi.cuda(), 3, 1, pad, 1).cpu().double(), want)


def test_wgrad_batched_matches_single_launches():
    """n weight gradients against one input in one launch (sgv3d_conv2d_backward_weight_batched) and the autograd function built
    on it (conv_grad.multi_conv2d): against the per-layer launches / float64 autograd."""
    g = torch.Generator().manual_seed(11)
    B, H, W, cin, cout, n = 2, 21, 45, 64, 64, 5
    x = torch.randn(B, H, W, cin, generator=g)
    dys = [torch.randn(B, H, W, cout, generator=g) for _ in range(n)]
    got = conv_grad.conv2d_backward_weight_batched(x.cuda(), [d.cuda() for d in dys])
    for d, dw in zip(dys, got):
        _, _, dw_ref = _reference(x, torch.zeros(cout, cin, 3, 3), d, 1, 1, 1)
        assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
    again = conv_grad.conv2d_backward_weight_batched(x.cuda(), [d.cuda() for d in dys])
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    # the autograd function: outputs, dx (sum over the layers) and every dw
    ws = [(torch.randn(cout, cin, 3, 3, generator=g) / 24).cuda().requires_grad_(True) for _ in range(n)]
    xg = x.cuda().requires_grad_(True)
    outs = conv_grad.multi_conv2d(xg, ws)
    loss = sum((o * d.cuda()).sum() for o, d in zip(outs, dys))
    loss.backward()
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = [w.detach().cpu().double().requires_grad_(True) for w in ws]
    lr = sum((F.conv2d(xr, w, None, 1, 1) * d.double().permute(0, 3, 1, 2)).sum() for w, d in zip(wr, dys))
    lr.backward()
    assert float((xg.grad.cpu().double().permute(0, 3, 1, 2) - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())
    for w, r in zip(ws, wr):
        assert float((w.grad.cpu().double() - r.grad).abs().max()) <= 2e-5 * float(r.grad.abs().max())


def test_wgrad_bf16_batched_matches_single_launches():
    """Mixed-precision mode: conv2d_backward_weight_batched runs the n gradients on the bf16 matrix cores in one launch
    (sgv3d_conv2d_backward_weight_bf16_batched); per problem it equals the single-layer bf16 launch to f32 summation order and the
    float64 gradient of the rounded operands to 2e-5; repeatable."""
    from sgv3d_amd import hip_ops
    g = torch.Generator().manual_seed(12)
    B, H, W, cin, cout, n = 2, 21, 45, 64, 64, 5
    x = torch.randn(B, H, W, cin, generator=g)
    dys = [torch.randn(B, H, W, cout, generator=g) for _ in range(n)]
    saved = hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS = True, False, False     # the per-tap form (all taps: its own test below)
    try:
        hip_ops.PROFILE = []
        got = conv_grad.conv2d_backward_weight_batched(x.cuda(), [d.cuda() for d in dys])
        assert [r[0] for r in hip_ops.PROFILE] == ["conv_wgrad_bf16"]
        hip_ops.PROFILE = None
        again = conv_grad.conv2d_backward_weight_batched(x.cuda(), [d.cuda() for d in dys])
    finally:
        hip_ops.PROFILE = None
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS = saved
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    for d, dw in zip(dys, got):
        _, _, dw_ref = _reference(x.bfloat16().float(), torch.zeros(cout, cin, 3, 3), d.bfloat16().float(), 1, 1, 1)
        assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
        single = conv_grad.conv2d_backward_weight_bf16(x.cuda(), d.cuda(), 3, 1, 1, 1, tile=1, split=3)
        assert float((dw - single).abs().max()) <= 2e-5 * float(dw_ref.abs().max())


def test_wgrad_is_exact_on_small_integers():
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (2, 30, 34, 64), generator=g).float()
    dy = torch.randint(-2, 3, (2, 30, 34, 128), generator=g).float()
    _, _, dw_ref = _reference(x, torch.zeros(128, 64, 3, 3), dy, 1, 1, 1)
    dw = conv_grad.conv2d_backward_weight(x.cuda(), dy.cuda(), 3, 1, 1, 1)
    assert torch.equal(dw.cpu().double(), dw_ref)


def test_transposed_conv_weight_gradient_by_role_swap():
    """ConvTranspose2d(kernel == stride): dW[ci][co][ky][kx] = wgrad of the equivalent strided conv with x and dy swapped."""
    g = torch.Generator().manual_seed(2)
    B, cin, cout, H, W, k = 1, 64, 32, 10, 12, 2
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cin, cout, k, k, generator=g)
    dy = torch.randn(B, H * k, W * k, cout, generator=g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    F.conv_transpose2d(xr, wr, stride=k).backward(dy.double().permute(0, 3, 1, 2))
    dw = conv_grad.conv2d_backward_weight(dy.cuda(), x.cuda(), k, k, 0, 1)       # roles swapped
    assert float((dw.cpu().double() - wr.grad).abs().max()) <= 2e-5 * float(wr.grad.abs().max())


def test_stem_weight_gradient_with_padded_image_channels():
    """The image stem: 3 weight channels over a 4-channel (zero-padded) NHWC image, taps packed into the tile columns."""
    g = torch.Generator().manual_seed(4)
    x3 = torch.randn(2, 45, 61, 3, generator=g)
    x4 = torch.cat([x3, torch.zeros(2, 45, 61, 1)], -1).contiguous()
    dy = torch.randn(2, 23, 31, 64, generator=g)
    _, _, dw_ref = _reference(x3, torch.zeros(64, 3, 7, 7), dy, 2, 3, 1)
    for split in (0, 1, 3):
        dw = conv_grad.conv2d_backward_weight(x4.cuda(), dy.cuda(), 7, 2, 3, 1, cin=3, split=split)
        assert dw.shape == (64, 3, 7, 7)
        assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())


def test_bias_gradient_and_chain():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 14, 18, 32, generator=g)
    w1 = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    b1 = torch.randn(64, generator=g)
    w2 = torch.randn(16, 64, 1, 1, generator=g) * 0.1
    params = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2)]
    out = conv_grad.conv2d(torch.relu(conv_grad.conv2d(params[0], params[1], params[2], 1, 1, 1)), params[3])
    out.square().sum().backward()
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2)]
    r = F.conv2d(torch.relu(F.conv2d(ref[0].permute(0, 3, 1, 2), ref[1], ref[2], 1, 1)), ref[3])
    r.square().sum().backward()
    for got, want in zip(params, ref):
        wg = want.grad
        assert float((got.grad.cpu().double() - wg).abs().max()) <= 5e-5 * float(wg.abs().max())


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, pad, couts
    (2, 64, 20, 28, 1, (2, 1, 3, 2, 2, 1)),       # one CenterHead task: reg, height, dim, rot, vel, heatmap
    (1, 64, 9, 75, 1, (4, 3, 3)),                 # rows of five 16-pixel / three 32-pixel segments with ragged ends, 4 channels
    (3, 24, 7, 17, 0, (1, 2)),                    # cin below one 64-channel chunk, no padding
    (1, 132, 6, 33, 2, (3, 4, 1, 2)),             # three channel chunks (the last one 4 channels), padding 2
])
def test_thin_backward_batched(shape):
    """The whole backward of n thin 3x3 layers in one call (sgv3d_conv3x3_thin_backward_batched): data, weight and bias gradients
    against float64 autograd, bitwise repeatable, exact on small integers; and each gradient kind on its own."""
    B, cin, H, W, pad, couts = shape
    g = torch.Generator().manual_seed(sum(shape[:5]) + len(couts))
    n = len(couts)
    oh, ow = H + 2 * pad - 2, W + 2 * pad - 2
    xs = [torch.randn(B, H, W, cin, generator=g) for _ in range(n)]
    dys = [torch.randn(B, oh, ow, c, generator=g) for c in couts]
    ws = [torch.randn(c, cin, 3, 3, generator=g) / 8 for c in couts]
    cu = lambda ts: [t.cuda() for t in ts]
    dxs, dws, dbs = conv_grad.thin_conv3x3_backward_batched(cu(xs), cu(dys), cu(ws), pad, need_dx=True, need_dw=True, need_db=True)
    for x, dy, w, dx, dw, db in zip(xs, dys, ws, dxs, dws, dbs):
        _, dx_ref, dw_ref = _reference(x, w, dy, 1, pad, 1)
        assert float((dx.cpu().double() - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max())
        assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
        db_ref = dy.double().sum((0, 1, 2))
        assert float((db.cpu().double() - db_ref).abs().max()) <= 2e-5 * max(1.0, float(db_ref.abs().max()))
    again = conv_grad.thin_conv3x3_backward_batched(cu(xs), cu(dys), cu(ws), pad, need_dx=True, need_dw=True, need_db=True)
    for a, b in zip(dxs + dws + dbs, again[0] + again[1] + again[2]):
        assert torch.equal(a, b)
    only_x = conv_grad.thin_conv3x3_backward_batched(cu(xs), cu(dys), cu(ws), pad, need_dx=True, need_dw=False, need_db=False)
    assert only_x[1] is None and only_x[2] is None and all(torch.equal(a, b) for a, b in zip(only_x[0], dxs))
    only_b = conv_grad.thin_conv3x3_backward_batched(cu(xs), cu(dys), cu(ws), pad, need_dx=False, need_dw=False, need_db=True)
    assert only_b[0] is None and only_b[1] is None and all(torch.equal(a, b) for a, b in zip(only_b[2], dbs))
    xi = [torch.randint(-3, 4, (B, H, W, cin), generator=g).float() for _ in range(n)]
    dyi = [torch.randint(-2, 3, (B, oh, ow, c), generator=g).float() for c in couts]
    wi = [torch.randint(-2, 3, (c, cin, 3, 3), generator=g).float() for c in couts]
    dxs, dws, dbs = conv_grad.thin_conv3x3_backward_batched(cu(xi), cu(dyi), cu(wi), pad, need_dx=True, need_dw=True, need_db=True)
    for x, dy, w, dx, dw, db in zip(xi, dyi, wi, dxs, dws, dbs):
        _, dx_ref, dw_ref = _reference(x, w, dy, 1, pad, 1)
        assert torch.equal(dx.cpu().double(), dx_ref) and torch.equal(dw.cpu().double(), dw_ref)
        assert torch.equal(db.cpu().double(), dy.double().sum((0, 1, 2)))


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, pad, couts
    (2, 64, 20, 28, 1, (2, 1, 3, 2, 2, 1)),
    (1, 64, 9, 75, 1, (4, 3, 3)),
    (3, 24, 7, 17, 0, (1, 2)),
    (1, 60, 6, 33, 2, (3, 4, 1, 2)),
])
def test_thin_forward_batched(shape):
    """sgv3d_conv3x3_thin_forward_batched (f32 FMAs + a 16-lane reduction per pixel): against float64, bitwise repeatable, exact on
    small integers, with and without bias."""
    B, cin, H, W, pad, couts = shape
    g = torch.Generator().manual_seed(sum(shape[:5]) + 3 * len(couts))
    xs = [torch.randn(B, H, W, cin, generator=g) for _ in couts]
    ws = [torch.randn(c, cin, 3, 3, generator=g) / 8 for c in couts]
    bs = [torch.randn(c, generator=g) if i % 2 == 0 else None for i, c in enumerate(couts)]
    cu = lambda ts: [None if t is None else t.cuda() for t in ts]
    ys = conv_grad.thin_conv3x3_forward_batched(cu(xs), cu(ws), cu(bs), pad)
    for x, w, b, y in zip(xs, ws, bs, ys):
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None if b is None else b.double(), 1, pad).permute(0, 2, 3, 1)
        assert tuple(y.shape) == tuple(ref.shape)
        assert float((y.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    again = conv_grad.thin_conv3x3_forward_batched(cu(xs), cu(ws), cu(bs), pad)
    assert all(torch.equal(a, b) for a, b in zip(ys, again))
    xi = [torch.randint(-3, 4, (B, H, W, cin), generator=g).float() for _ in couts]
    wi = [torch.randint(-2, 3, (c, cin, 3, 3), generator=g).float() for c in couts]
    for x, w, y in zip(xi, wi, conv_grad.thin_conv3x3_forward_batched(cu(xi), cu(wi), None, pad)):
        assert torch.equal(y.cpu().double(), F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, 1, pad).permute(0, 2, 3, 1))


def test_multi_thin_conv2d_matches_the_per_layer_function():
    """conv_grad.multi_thin_conv2d (what the CenterHead's final layers use in training): outputs and gradients of inputs / weights /
    biases against the per-layer autograd function (float32 summation order apart)."""
    import torch.nn as nn
    torch.manual_seed(5)
    B, H, W, cin = 2, 24, 40, 64
    couts = (2, 1, 3, 2)
    convs = [nn.Conv2d(cin, c, 3, padding=1).cuda() for c in couts]
    xs = [torch.randn(B, H, W, cin, device='cuda').requires_grad_(True) for _ in couts]
    ups = [torch.randn(B, H, W, c, device='cuda') for c in couts]
    assert conv_grad.thin_conv_eligible(convs, xs)
    outs = conv_grad.multi_thin_conv2d(xs, convs)
    sum((o * u).sum() for o, u in zip(outs, ups)).backward()
    got = [(o.detach(), x.grad.clone(), c.weight.grad.clone(), c.bias.grad.clone()) for o, x, c in zip(outs, xs, convs)]
    for x, c in zip(xs, convs):
        x.grad = None; c.weight.grad = None; c.bias.grad = None
    ref = [conv_grad.conv2d(x, c.weight, c.bias, 1, 1, 1) for x, c in zip(xs, convs)]
    sum((o * u).sum() for o, u in zip(ref, ups)).backward()
    for (o, dx, dw, db), r, x, c in zip(got, ref, xs, convs):
        assert float((o - r.detach()).abs().max()) <= 3e-5 * float(r.abs().max())          # (f32 FMAs here, MFMA summation order there)
        for a, b in ((dx, x.grad), (dw, c.weight.grad), (db, c.bias.grad)):
            assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()), (a.shape, float((a - b).abs().max()), float(b.abs().max()))
    assert not conv_grad.thin_conv_eligible(convs + [nn.Conv2d(cin, 8, 3, padding=1).cuda()], xs + [xs[0]])


@pytest.mark.parametrize("shape", [
    # B, cin, H, W, cout, pad, dil
    (2, 64, 20, 28, 64, 1, 1),           # one tile, a ragged 32-pixel segment
    (1, 128, 19, 70, 64, 1, 1),          # three segments (the last one 6 pixels), two input-channel tiles
    (2, 96, 17, 33, 160, 1, 1),          # channel counts that are not multiples of 64 (partial tiles on both sides)
    (1, 64, 23, 40, 64, 6, 6),           # ASPP dilation 6: six interleaved row phases, staged rows of 44 pixels
    (1, 64, 21, 36, 128, 18, 18),        # dilation 18 = the map is smaller than the dilated kernel's reach in places
    (1, 72, 12, 16, 68, 0, 1),           # no padding, 4-channel remainders
    (2, 64, 9, 31, 64, 2, 1),            # padding 2 (output larger than the input)
])
def test_wgrad_bf16_alltaps(shape):
    """sgv3d_conv2d_backward_weight_bf16_alltaps (tile 6: all nine taps of a 3x3 / stride-1 layer per workgroup, operands read from
    NHWC-ordered LDS images with the transposing read): against the float64 gradient of the bf16-rounded operands, exact on small
    integers, bitwise repeatable, every chunk count the same up to summation order, and channel windows of wider tensors."""
    B, cin, H, W, cout, pad, dil = shape
    g = torch.Generator().manual_seed(sum(shape))
    oh, ow = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
    x = torch.randn(B, H, W, cin + 8, generator=g)
    dy = torch.randn(B, oh, ow, cout + 4, generator=g)
    rb = lambda t: t.bfloat16().float()
    _, _, dw_ref = _reference(rb(x[..., 4:4 + cin]), torch.zeros(cout, cin, 3, 3), rb(dy[..., 4:4 + cout]), 1, pad, dil)
    run = lambda split: conv_grad.conv2d_backward_weight_bf16(x.cuda(), dy.cuda(), 3, 1, pad, dil, cin=cin, cout=cout, x_coff=4, y_coff=4,
                                                              tile=6, split=split)
    dw = run(0)
    scale = float(dw_ref.abs().max())
    assert float((dw.cpu().double() - dw_ref).abs().max()) <= 2e-5 * scale
    assert torch.equal(dw, run(0))
    for split in (1, 2, 3):
        assert float((run(split) - dw).abs().max()) <= 2e-5 * scale
    per_tap = conv_grad.conv2d_backward_weight_bf16(x.cuda(), dy.cuda(), 3, 1, pad, dil, cin=cin, cout=cout, x_coff=4, y_coff=4, tile=1, split=2)
    assert float((per_tap - dw).abs().max()) <= 2e-5 * scale
    xi = torch.randint(-3, 4, (B, H, W, cin), generator=g).float()
    dyi = torch.randint(-2, 3, (B, oh, ow, cout), generator=g).float()
    _, _, want = _reference(xi, torch.zeros(cout, cin, 3, 3), dyi, 1, pad, dil)
    assert torch.equal(conv_grad.conv2d_backward_weight_bf16(xi.cuda(), dyi.cuda(), 3, 1, pad, dil, tile=6).cpu().double(), want)


def test_wgrad_bf16_alltaps_batched():
    """The batched form (n gradients against one input, what the 36 first layers of the CenterHead branches use in the mixed-precision
    step): each problem bitwise the single-problem launch with the same chunking rule applied to n = 1 ... here compared to 2e-5."""
    from sgv3d_amd import hip_ops
    g = torch.Generator().manual_seed(21)
    B, H, W, cin, cout, n = 2, 21, 45, 64, 64, 5
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    dys = [torch.randn(B, H, W, cout, generator=g).cuda() for _ in range(n)]
    saved = (hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS)
    try:
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS = True, False, True
        got = conv_grad.conv2d_backward_weight_batched(x, dys)
        again = conv_grad.conv2d_backward_weight_batched(x, dys)
        hip_ops.WGRAD_BF16_ALLTAPS = False
        per_tap = conv_grad.conv2d_backward_weight_batched(x, dys)
    finally:
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS = saved
    rb = lambda t: t.cpu().bfloat16().float()
    for d, dw, dw2, dw3 in zip(dys, got, again, per_tap):
        _, _, ref = _reference(rb(x), torch.zeros(cout, cin, 3, 3), rb(d), 1, 1, 1)
        scale = float(ref.abs().max())
        assert float((dw.cpu().double() - ref).abs().max()) <= 2e-5 * scale
        assert torch.equal(dw, dw2)
        assert float((dw - dw3).abs().max()) <= 2e-5 * scale
