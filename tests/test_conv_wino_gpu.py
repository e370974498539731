"""Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) through the C ABI vs torch conv2d and vs the implicit GEMM."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(B, cin, H, W, cout, seed=0, integer=False):
    g = torch.Generator().manual_seed(seed)
    if integer:
        x = torch.randint(-3, 4, (B, H, W, cin), generator=g).float()
        w = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).float()
    else:
        x = torch.randn(B, H, W, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    return x, w


def _ref(x, w, scale=None, shift=None, residual=None, relu=False):
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    if scale is not None:
        y = y * scale.double() + shift.double()
    if residual is not None:
        y = y + residual.double()
    if relu:
        y = y.clamp_min(0)
    return y


SHAPES = [
    (1, 8, 16, 16, 64),      # one block, one step
    (1, 64, 54, 96, 64),     # ragged blocks (54 = 3*16 + 6)
    (2, 32, 17, 33, 96),     # batch 2, odd sizes, cout not a multiple of 64
    (1, 160, 32, 32, 160),   # cout = 2.5 tiles
    (1, 512, 27, 48, 512),
    (1, 64, 37, 21, 20),     # tiny cout
    (1, 16, 8, 32, 64),      # exactly one 8x32-pixel block (4 x 16 tiles)
    (2, 32, 20, 70, 96),     # 4 x 16-tile blocks with ragged right / bottom edges, batch 2
]


VARIANTS = [5, 8]       # hip_ops.TILE_WINO (one workgroup per CU), TILE_WINO_HALF (positions split over wave pairs)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("TILE_WINO", VARIANTS)
def test_winograd_matches_conv2d(shape, split, TILE_WINO):
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = shape
    if cin // 8 < split:
        pytest.skip("not enough k-steps")
    x, w = _mk(B, cin, H, W, cout)
    conv = PackedConv(w.cuda(), pad=1)
    assert conv.w_wino is not None
    y = conv(x.cuda(), tile=TILE_WINO, split_k=split)
    ref = _ref(x, w)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("TILE_WINO", VARIANTS)
def test_winograd_bit_exact_on_integer_data(TILE_WINO):
    from sgv3d_amd.hip_ops import PackedConv
    x, w = _mk(2, 32, 20, 36, 72, integer=True)
    conv = PackedConv(w.cuda(), pad=1)
    y = conv(x.cuda(), tile=TILE_WINO, split_k=1)
    assert torch.equal(y.cpu().double(), _ref(x, w))
    y2 = conv(x.cuda(), tile=TILE_WINO, split_k=2)
    assert torch.equal(y2.cpu().double(), _ref(x, w))


@pytest.mark.parametrize("TILE_WINO", VARIANTS)
def test_winograd_epilogue_bn_residual_relu_gate_and_slices(TILE_WINO):
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = 2, 32, 19, 23, 48
    x, w = _mk(B, cin, H, W, cout, seed=3)
    g = torch.Generator().manual_seed(4)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(B, H, W, cout, generator=g)
    gate = torch.rand(B, cout, generator=g)
    conv = PackedConv(w.cuda(), pad=1, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    # input read from a channel slice of a wider buffer, output written into a slice of a wider buffer
    xw = torch.randn(B, H, W, cin + 16, generator=g)
    xw[..., 8:8 + cin] = x
    out = torch.full((B, H, W, cout + 8), -7.0).cuda()
    for split in (1, 2):
        conv(xw.cuda(), out, x_coff=8, y_coff=4, residual=res.cuda(), gate=gate.cuda(), tile=TILE_WINO, split_k=split)
        ref = _ref(x, w, scale, shift, res, relu=True) * gate.double()[:, None, None, :]
        got = out.cpu().double()
        assert (got[..., 4:4 + cout] - ref).abs().max().item() < 1e-4
        assert (got[..., :4] == -7).all() and (got[..., 4 + cout:] == -7).all()


@pytest.mark.parametrize("TILE_WINO", VARIANTS)
def test_winograd_modes_match_implicit_gemm(TILE_WINO):
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = 1, 64, 40, 24, 128
    x, w = _mk(B, cin, H, W, cout, seed=5)
    conv = PackedConv(w.cuda(), pad=1, relu=True)
    xc = x.cuda()
    a = conv(xc, nchw_out=True, tile=4, split_k=1)
    b = conv(xc, nchw_out=True, tile=TILE_WINO, split_k=1)
    assert (a - b).abs().max().item() < 1e-4
    a = conv(xc, group_planes=64, tile=4, split_k=1)
    b = conv(xc, group_planes=64, tile=TILE_WINO, split_k=1)
    assert a.shape == b.shape and (a - b).abs().max().item() < 1e-4


def test_winograd_rejects_unsupported_layers():
    from sgv3d_amd import _lib
    from sgv3d_amd.hip_ops import PackedConv, TILE_WINO
    x, w = _mk(1, 16, 12, 12, 8)
    conv = PackedConv(w.cuda(), stride=2, pad=1)
    assert conv.w_wino is None
    with pytest.raises(_lib.SGV3DError):
        conv(x.cuda(), tile=TILE_WINO, split_k=1)


def test_winograd_deterministic():
    from sgv3d_amd.hip_ops import PackedConv, TILE_WINO
    x, w = _mk(1, 128, 54, 96, 128, seed=9)
    conv = PackedConv(w.cuda(), pad=1)
    xc = x.cuda()
    a = conv(xc, tile=TILE_WINO, split_k=3).clone()
    for _ in range(5):
        assert torch.equal(a, conv(xc, tile=TILE_WINO, split_k=3))


@pytest.mark.parametrize("shape", [(1, 64, 40, 56, 256), (2, 32, 17, 33, 200), (1, 96, 16, 16, 128), (1, 8, 70, 20, 64)])
def test_winograd_patch_resident_variant(shape):
    """cin <= 96: the whole patch stays in LDS and one workgroup walks over the cout tiles."""
    from sgv3d_amd.hip_ops import PackedConv, TILE_WINO, TILE_WINO_RES
    B, cin, H, W, cout = shape
    x, w = _mk(B, cin, H, W, cout, seed=11)
    g = torch.Generator().manual_seed(12)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    conv = PackedConv(w.cuda(), pad=1, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    xc = x.cuda()
    y = conv(xc, tile=TILE_WINO_RES, split_k=1)
    ref = _ref(x, w, scale, shift, relu=True)
    assert (y.cpu().double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    assert torch.equal(y, conv(xc, tile=TILE_WINO, split_k=1))        # same arithmetic, same order
    if cout % 64 == 0:
        a = conv(xc, group_planes=64, tile=TILE_WINO_RES, split_k=1)
        b = conv(xc, group_planes=64, tile=TILE_WINO, split_k=1)
        assert torch.equal(a, b)


def test_winograd_patch_resident_rejects_wide_inputs():
    from sgv3d_amd import _lib
    from sgv3d_amd.hip_ops import PackedConv, TILE_WINO_RES
    x, w = _mk(1, 128, 16, 16, 128)
    conv = PackedConv(w.cuda(), pad=1)
    with pytest.raises(_lib.SGV3DError):
        conv(x.cuda(), tile=TILE_WINO_RES, split_k=1)


@pytest.mark.parametrize("B,H,W,cin,counts", [(1, 32, 48, 64, (2, 1, 3, 2)), (2, 20, 37, 32, (1, 3)), (1, 16, 16, 64, (4,)),
                                             (1, 50, 33, 64, (2, 1, 3, 2, 2, 1, 1, 3))])
def test_fused_centerhead_branches(B, H, W, cin, counts):
    """sgv3d_centerhead_branches_forward == [3x3 cin->64 + BN + ReLU] then [3x3 64->c + bias] per branch."""
    from sgv3d_amd import hip_ops
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(21)
    nb = len(counts)
    x = torch.randn(B, H, W, cin, generator=g)
    w1 = torch.randn(nb * 64, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    sc, sh = torch.rand(nb * 64, generator=g) + 0.5, torch.randn(nb * 64, generator=g) * 0.2
    total = sum(counts)
    w2 = torch.randn(total, 64, 3, 3, generator=g) / 24.0
    b2 = torch.randn(total, generator=g)
    hid = (F.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), padding=1) * sc.double()[None, :, None, None]
           + sh.double()[None, :, None, None]).clamp_min(0)
    ref, off = [], 0
    for k, c in enumerate(counts):
        ref.append(F.conv2d(hid[:, k * 64:(k + 1) * 64], w2[off:off + c].double(), b2[off:off + c].double(), padding=1))
        off += c
    ref = torch.cat(ref, 1)
    first = PackedConv(w1.cuda(), pad=1, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    out = hip_ops.centerhead_branches(x.cuda(), first, w2.permute(0, 2, 3, 1).contiguous().cuda(), b2.cuda(), ob, nb)
    assert out.shape == ref.shape
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    again = hip_ops.centerhead_branches(x.cuda(), first, w2.permute(0, 2, 3, 1).contiguous().cuda(), b2.cuda(), ob, nb)
    assert torch.equal(out, again)          # fixed summation order: bit-reproducible


@pytest.mark.parametrize("B,H,W,counts", [(1, 32, 48, (2, 1, 3, 2)), (2, 20, 37, (1, 3)), (1, 16, 16, (4,)),
                                         (1, 50, 33, (2, 1, 3, 2, 2, 1, 1, 3)), (1, 64, 64, (5, 1, 7))])
def test_fused_centerhead_branches_f4(B, H, W, counts):
    """sgv3d_centerhead_branches_forward_f4 (F(4x4) first layers, V resident in LDS, scatter-form final convolution) ==
    [3x3 64->64 + BN + ReLU] then [3x3 64->c + bias] per branch; ragged blocks, two images, branches wider than 4."""
    from sgv3d_amd import hip_ops
    g = torch.Generator().manual_seed(44)
    nb, cin = len(counts), 64
    x = torch.randn(B, H, W, cin, generator=g)
    w1 = torch.randn(nb * 64, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    sc, sh = torch.rand(nb * 64, generator=g) + 0.5, torch.randn(nb * 64, generator=g) * 0.2
    total = sum(counts)
    w2 = torch.randn(total, 64, 3, 3, generator=g) / 24.0
    b2 = torch.randn(total, generator=g)
    hid = (F.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), padding=1) * sc.double()[None, :, None, None]
           + sh.double()[None, :, None, None]).clamp_min(0)
    ref, off = [], 0
    for k, c in enumerate(counts):
        ref.append(F.conv2d(hid[:, k * 64:(k + 1) * 64], w2[off:off + c].double(), b2[off:off + c].double(), padding=1))
        off += c
    ref = torch.cat(ref, 1)
    u = hip_ops.pack_centerhead_f4(w1.cuda())
    ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    args = (x.cuda(), u, sc.cuda(), sh.cuda(), w2.permute(0, 2, 3, 1).contiguous().cuda(), b2.cuda(), ob, nb)
    out = hip_ops.centerhead_branches_f4(*args)
    assert out.shape == ref.shape
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err
    assert torch.equal(out, hip_ops.centerhead_branches_f4(*args))          # fixed summation order: bit-reproducible
    # a wider buffer (the shared map as a channel slice) gives the same bits
    wide = torch.cat([x, torch.randn(B, H, W, 16, generator=g)], -1).cuda()
    assert torch.equal(out, hip_ops.centerhead_branches_f4(wide, *args[1:]))


def test_fused_centerhead_f4_exact_on_small_integers():
    """Integer inputs and weights chosen so that every intermediate of the F(4x4) path is exactly representable: the
    kernel's index arithmetic (tile order, swizzle, DPP overlap-add, lane folds, ring) is then checked bit for bit."""
    from sgv3d_amd import hip_ops
    g = torch.Generator().manual_seed(5)
    B, H, W, counts = 1, 40, 24, (2, 3)
    nb, total = len(counts), sum(counts)
    x = torch.randint(-2, 3, (B, H, W, 64), generator=g).float()
    # G g G^T is exact in binary floating point when the taps are multiples of 576 = 24^2 (G's entries are k / 24)
    w1 = torch.randint(-1, 2, (nb * 64, 64, 3, 3), generator=g).float() * 576.0
    w1 = w1 * (torch.rand(nb * 64, 64, 1, 1, generator=g) < 0.08)           # sparse: sums stay far below 2^24
    sc = torch.full((nb * 64,), 1.0 / 64.0)                                 # a power of two: hidden = 9 k + shift, exactly
    sh = torch.randint(-2, 3, (nb * 64,), generator=g).float()
    w2 = torch.randint(-2, 3, (total, 64, 3, 3), generator=g).float()
    b2 = torch.randint(-3, 4, (total,), generator=g).float()
    hid = (F.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), padding=1) * sc.double()[None, :, None, None]
           + sh.double()[None, :, None, None]).clamp_min(0)
    ref, off = [], 0
    for k, c in enumerate(counts):
        ref.append(F.conv2d(hid[:, k * 64:(k + 1) * 64], w2[off:off + c].double(), b2[off:off + c].double(), padding=1))
        off += c
    ref = torch.cat(ref, 1)
    ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    out = hip_ops.centerhead_branches_f4(x.cuda(), hip_ops.pack_centerhead_f4(w1.cuda()), sc.cuda(), sh.cuda(),
                                         w2.permute(0, 2, 3, 1).contiguous().cuda(), b2.cuda(), ob, nb)
    assert torch.equal(out.cpu().double(), ref)


F4RES_SHAPES = [
    (1, 64, 16, 16, 64, 64),      # one block
    (1, 64, 54, 96, 64, 64),      # ResNet layer 1 at quarter size: ragged bottom blocks
    (2, 64, 21, 37, 192, 64),     # batch 2, three output tiles, odd sizes
    (1, 256, 48, 40, 64, 256),    # the CenterHead's shared layer: four input chunks
    (1, 128, 20, 20, 64, 120),    # real cin below the padded one (zero weight columns)
]


@pytest.mark.parametrize("shape", F4RES_SHAPES)
@pytest.mark.parametrize("epi", ["plain", "bn_relu", "residual_relu"])
def test_winograd4_resident_matches_conv2d(shape, epi):
    """sgv3d_conv3x3_f4res_forward (conv_f4res_kernel: F(4x4) with V resident in LDS) through PackedConv(tile=TILE_F4RES) vs the
    float64 definition; output written into a channel slice of a wider buffer, input read from one."""
    from sgv3d_amd import hip_ops
    from sgv3d_amd.hip_ops import PackedConv, TILE_F4RES
    B, cin, H, W, cout, cin_real = shape
    x, w = _mk(B, cin, H, W, cout, seed=3)
    w = w[:, :cin_real].contiguous()
    x[..., cin_real:] = 7.0                                       # must be multiplied by zero weights
    g = torch.Generator().manual_seed(9)
    scale = shift = res = None
    if epi != "plain":
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    if epi == "residual_relu":
        res = torch.randn(B, H, W, cout, generator=g)
    conv = PackedConv(w.cuda(), pad=1, cin_pad=cin, scale=None if scale is None else scale.cuda(),
                      shift=None if shift is None else shift.cuda(), relu=epi != "plain")
    assert conv.f4res_ok()
    xin = torch.cat([torch.full((B, H, W, 8), -3.0), x], -1).cuda()            # channels [8, 8 + cin) of a wider map
    out = torch.full((B, H, W, cout + 12), 5.0, device="cuda")
    y = conv(xin, out=out, x_coff=8, y_coff=4, residual=None if res is None else res.cuda(), tile=TILE_F4RES, split_k=1)
    ref = _ref(x[..., :cin_real], w, scale, shift, res, relu=epi != "plain")
    got = y[..., 4:4 + cout].cpu().double()
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err
    assert bool((y[..., :4] == 5.0).all()) and bool((y[..., 4 + cout:] == 5.0).all())      # neighbours of the slice untouched
    again = conv(xin, out=torch.full_like(out, 5.0), x_coff=8, y_coff=4, residual=None if res is None else res.cuda(),
                 tile=TILE_F4RES, split_k=1)
    assert torch.equal(y, again)


def test_winograd4_resident_rejects_other_shapes():
    from sgv3d_amd import _lib
    from sgv3d_amd.hip_ops import PackedConv, TILE_F4RES
    x, w = _mk(1, 128, 16, 16, 128)
    conv = PackedConv(w.cuda(), pad=1)
    assert not conv.f4res_ok()
    with pytest.raises(_lib.SGV3DError):
        conv(x.cuda(), tile=TILE_F4RES, split_k=1)


# ------------------------------------------------------------------------------------------------ F(4x4, 3x3)
WINO4_SHAPES = [
    (1, 128, 8, 8, 128),       # exactly 2 x 2 tiles
    (1, 512, 54, 96, 512),     # the HeightNet layer of cfg-2: 14 x 24 tiles (54 = 13 * 4 + 2), 336 -> 384 rows per position
    (2, 128, 17, 33, 160),     # batch 2, ragged bottom / right tiles, cout = 1.25 GEMM tiles
    (1, 256, 27, 48, 256),     # ResNet layer 3 at half size
    (1, 640, 32, 32, 640),     # BEV trunk, last stage
]


@pytest.mark.parametrize("shape", WINO4_SHAPES)
@pytest.mark.parametrize("tile", [9, 10, 15, 46, 47] + list(range(50, 60)))  # hip_ops.TILE_WINO4 (64x64 GEMM tile), _WIDE (64x128), _NARROW (32x128), _OCC (64x64, five per CU), _G48 (16x16x4 MFMA, 48x64); 50-59: the f32x3 position GEMM (bf16 matrix cores, 48 / 64 / 96 / 112 / 128 rows x 128 / 160 columns)
def test_winograd_f4x4_matches_conv2d(shape, tile):
    """csrc/conv_wino4.hip: input transform -> grouped GEMM over the 36 positions -> output transform, with folded BN,
    residual, ReLU and a concat offset, against a float64 convolution.  fp32 bound: 1e-4 of the output scale (measured ~1e-5:
    the transformed operands are up to 100x the inputs)."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = shape
    x, w = _mk(B, cin, H, W, cout, seed=4)
    g = torch.Generator().manual_seed(44)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, H, W, cout, generator=g)
    conv = PackedConv(w.cuda(), pad=1, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    assert conv.wino4_ok()
    out = torch.full((B, H, W, cout + 8), -3.0, device="cuda")
    conv(x.cuda(), out, y_coff=4, residual=res.cuda(), tile=tile, split_k=1)
    ref = _ref(x, w, scale, shift, res, relu=True)
    err = (out[..., 4:cout + 4].cpu().double() - ref).abs().max().item()
    print(f"F(4x4) {shape} tile {tile}: max err {err:.2e} (scale {ref.abs().max().item():.1f})")
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    assert float(out[..., :4].max()) == -3.0 and float(out[..., cout + 4:].max()) == -3.0     # neighbours untouched
    again = torch.full_like(out, -3.0)
    conv(x.cuda(), again, y_coff=4, residual=res.cuda(), tile=tile, split_k=1)
    assert torch.equal(out, again)                                                            # deterministic
    # and against the F(2x2) kernel on the same layer
    y2 = conv(x.cuda(), residual=res.cuda(), tile=5, split_k=1)
    assert (y2 - out[..., 4:cout + 4]).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_winograd_f4x4_rejects_what_it_does_not_cover():
    from sgv3d_amd._lib import SGV3DError
    from sgv3d_amd.hip_ops import PackedConv
    x, w = _mk(1, 64, 16, 16, 64)
    conv = PackedConv(w.cuda(), pad=1)
    assert not conv.wino4_ok()                                       # fewer channels than the transforms are worth
    with pytest.raises(SGV3DError):
        conv(x.cuda(), tile=9, split_k=1)
    x, w = _mk(1, 128, 16, 16, 128)
    conv = PackedConv(w.cuda(), pad=1)
    with pytest.raises(SGV3DError):
        conv(x.cuda(), tile=9, split_k=2)                            # no split-K


@pytest.mark.parametrize("tile", [9, 51, 53])
@pytest.mark.parametrize("dil,H,W", [(6, 54, 96), (12, 54, 96), (18, 54, 96), (2, 19, 23), (3, 20, 20)])
def test_winograd_f4x4_dilated(dil, H, W, tile):
    """Dilated 3x3 (pad == dilation: the ASPP branches): dil x dil independent sub-grid convolutions through the same three
    launches, tiles of 4x4 outputs spaced `dil` apart; sub-grids of different sizes (54 = 4 * 12 + 6), ragged tiles."""
    from sgv3d_amd.hip_ops import PackedConv
    cin = cout = 128 if H < 54 else 512
    g = torch.Generator().manual_seed(dil)
    x = torch.randn(1, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    conv = PackedConv(w.cuda(), pad=dil, dil=dil, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    assert conv.wino4_ok() and conv.w_wino is None
    out = torch.full((1, H, W, cout + 64), -3.0, device="cuda")
    conv(x.cuda(), out, y_coff=32, tile=tile, split_k=1)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=dil, dilation=dil).permute(0, 2, 3, 1)
    ref = (ref * scale.double() + shift.double()).clamp_min(0)
    err = (out[..., 32:cout + 32].cpu().double() - ref).abs().max().item()
    print(f"F(4x4) dilation {dil} @{H}x{W}: max err {err:.2e} (scale {ref.abs().max().item():.1f})")
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    assert float(out[..., :32].max()) == -3.0 and float(out[..., cout + 32:].max()) == -3.0
    direct = conv(x.cuda(), tile=4, split_k=1)
    assert (direct - out[..., 32:cout + 32]).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("tile", [9, 10, 52, 58])
def test_winograd_f4x4_group_planes_in_channel_chunks(tile):
    """The CenterHead first layers as one convolution (64 -> 36 x 64) with the hidden maps as [branch][B][H][W][64] planes:
    F(4x4) runs the GEMM + output transform in chunks of output channels (M of a pass stays below ~160 MB), the input
    transform once.  72 x 80 pixels, batch 2: 360 tiles -> chunks of 1664 + 640 channels."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, nb = 2, 64, 72, 80, 36
    x, w = _mk(B, cin, H, W, nb * 64, seed=9)
    g = torch.Generator().manual_seed(99)
    scale, shift = torch.rand(nb * 64, generator=g) + 0.5, torch.randn(nb * 64, generator=g) * 0.2
    conv = PackedConv(w.cuda(), pad=1, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    assert conv.wino4_ok()
    hidden = conv(x.cuda(), group_planes=64, tile=tile, split_k=1)
    assert tuple(hidden.shape) == (nb, B, H, W, 64)
    ref = _ref(x, w, scale, shift, None, relu=True)                              # [B, H, W, nb * 64]
    got = hidden.permute(1, 2, 3, 0, 4).reshape(B, H, W, nb * 64).cpu().double()
    err = (got - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    direct = conv(x.cuda(), group_planes=64, tile=4, split_k=1)
    assert (direct - hidden).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(1, 512, 54, 96, 512), (1, 160, 64, 64, 160), (2, 128, 17, 33, 160)])
def test_winograd_f4x4_x3_products_are_f32_products(shape):
    """The f32x3 position GEMM (csrc/gemm_x3_grouped.hip: every operand split exactly into three bf16 terms by its producer, six
    partial products accumulated in f32) against the f32-MFMA position GEMM of the same layer, both against float64: the x3 form is
    as close to float64 as the native one (bar: within 1.5x of its error; measured 0.6-0.9x -- the dropped terms are below one f32
    rounding of a product), on mixed-magnitude data (activations over four decades) where a bf16-only product would be off by 1e-2."""
    from sgv3d_amd.hip_ops import PackedConv
    B, cin, H, W, cout = shape
    x, w = _mk(B, cin, H, W, cout, seed=14)
    g = torch.Generator().manual_seed(15)
    x = x * torch.pow(10.0, torch.randint(-2, 3, (1, 1, 1, cin), generator=g).float())       # per-channel magnitudes 1e-2 .. 1e2
    conv = PackedConv(w.cuda(), pad=1)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    native = float((conv(x.cuda(), tile=47, split_k=1).cpu().double() - ref).abs().max()) / scale
    worst = 0.0
    for tile in [t for t in range(50, 60)]:
        y = conv(x.cuda(), tile=tile, split_k=1)
        worst = max(worst, float((y.cpu().double() - ref).abs().max()) / scale)
    print(f"{shape}: f32 MFMA {native:.2e}, f32x3 worst of ten tile shapes {worst:.2e} of the output scale")
    assert worst <= 1.5 * native and worst < 1e-4, (worst, native)


def test_winograd_f4x4_x3_keeps_every_partial_product():
    """Inputs in {-1, 0, 1} and weights that are multiples of 576 = 24^2 (G's denominators 4, 6, 24, squared): V is integer, U is
    integer up to the rounding of 1 / 6 in the packer, every partial sum of a position GEMM stays below 2^24 -- both GEMMs are then
    accurate to ~1e-7 of the output scale, and a dropped or doubled partial product (a missing lo term is 2^-16 = 1.5e-5 of a
    product), a swapped plane or a wrong k chunk stands far out of the 2e-6 bar, on every tile shape."""
    from sgv3d_amd.hip_ops import PackedConv
    g = torch.Generator().manual_seed(21)
    B, cin, H, W, cout = 1, 128, 24, 40, 160
    x = torch.randint(-1, 2, (B, H, W, cin), generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float() * 576.0
    conv = PackedConv(w.cuda(), pad=1)
    want = conv(x.cuda(), tile=47, split_k=1).cpu().double()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((want - ref).abs().max()) <= 2e-6 * scale
    for tile in range(50, 60):
        y = conv(x.cuda(), tile=tile, split_k=1).cpu().double()
        assert float((y - ref).abs().max()) <= 2e-6 * scale and float((y - want).abs().max()) <= 2e-6 * scale, tile
