"""GPU: parity at the FULL sizes of the BASELINE configs (the launches bench.py actually times).

* the two riskiest single launches of cfg-2 -- the 36-branch fused CenterHead kernel at 256x256 (256 workgroups + ring
  fix-up) and a HeightNet 512->512 3x3 layer at 54x96 with whatever (algorithm, tile, split-K) the autotuner picks --
  against a float64 convolution (im2col + dgemm on the device: a checker, not the product);
* the whole BEVHeight forward at cfg-2 (R50, 864x1536 -> 256x256), cfg-3 (R101, 1088x1920 -> 512x512) and cfg-5
  (SGV3D BSM R101, stride-8 frustum, D=180) size, one frame, against the torch-CPU oracle on the same weights:
  fp32 within the north_star's 1e-3 with bit-exact voxel indices, and the bf16-MFMA mode (the compute dtype cfg-3 /
  cfg-5 name) within the tolerance derived below.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_model as TM
from sgv3d_amd import hip_ops, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def conv3x3_f64(x, w, bias=None):
    """float64 3x3 / pad 1 convolution on the device as im2col + dgemm.  x [B,C,H,W], w [O,C,3,3] (any float dtype)."""
    B, C, H, W = x.shape
    cols = F.unfold(x.double(), 3, padding=1)                              # [B, C*9, H*W]
    y = w.double().reshape(w.shape[0], -1) @ cols                          # [B, O, H*W]
    if bias is not None:
        y = y + bias.double()[None, :, None]
    return y.reshape(B, w.shape[0], H, W)


def test_fused_centerhead_36_branches_256x256():
    """The cfg-2 launch: 6 tasks x (reg 2, height 1, dim 3, rot 2, vel 2, heatmap nc) on a 64-channel 256x256 map."""
    from sgv3d_amd.hip_ops import PackedConv
    counts = []
    for nc in (1, 2, 2, 1, 2, 2):
        counts += [2, 1, 3, 2, 2, nc]
    nb, total = len(counts), sum(counts)
    assert nb == 36 and total == 70
    g = torch.Generator().manual_seed(36)
    B, H, W, cin = 1, 256, 256, 64
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    w1 = (torch.randn(nb * 64, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(DEV)
    sc = (torch.rand(nb * 64, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(nb * 64, generator=g) * 0.2).to(DEV)
    w2 = (torch.randn(total, 64, 3, 3, generator=g) / 24.0).to(DEV)
    b2 = torch.randn(total, generator=g).to(DEV)
    first = PackedConv(w1, pad=1, scale=sc, shift=sh, relu=True)
    ob = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=DEV)
    out = hip_ops.centerhead_branches(x, first, w2.permute(0, 2, 3, 1).contiguous(), b2, ob, nb)
    again = hip_ops.centerhead_branches(x, first, w2.permute(0, 2, 3, 1).contiguous(), b2, ob, nb)
    assert torch.equal(out, again)                                          # fixed summation order
    xn = x.permute(0, 3, 1, 2)
    worst, scale, off = 0.0, 0.0, 0
    for k, c in enumerate(counts):                                          # one branch at a time: 300 MB of im2col each
        hid = conv3x3_f64(xn, w1[k * 64:(k + 1) * 64]) * sc[k * 64:(k + 1) * 64].double()[None, :, None, None] \
            + sh[k * 64:(k + 1) * 64].double()[None, :, None, None]
        ref = conv3x3_f64(hid.clamp_min(0), w2[off:off + c], b2[off:off + c])
        worst = max(worst, float((out[:, off:off + c].double() - ref).abs().max()))
        scale = max(scale, float(ref.abs().max()))
        off += c
    assert worst < 1e-4 * max(1.0, scale), (worst, scale)
    # and the two-kernel path (hidden maps through HBM) computes the same function
    hidden = first(x, group_planes=64)
    two = hip_ops.head_final_conv(hidden, w2.permute(0, 2, 3, 1).contiguous(), b2,
                                  torch.repeat_interleave(torch.arange(nb, dtype=torch.int32), torch.tensor(counts)).to(DEV),
                                  nb, 64)
    assert float((two - out).abs().max()) < 1e-4 * max(1.0, scale)


@pytest.mark.parametrize("with_residual", [False, True])
def test_heightnet_512_layer_54x96_autotuned(with_residual):
    """512->512 3x3 @54x96 (ten of them per cfg-2 frame), BN + ReLU (+ residual: the BasicBlock form), called the way
    the model calls it: (algorithm, tile, split-K) from the first-call measurement.  Every Winograd split the
    autotuner may pick is checked too."""
    from sgv3d_amd.hip_ops import PackedConv, TILE_WINO
    g = torch.Generator().manual_seed(512)
    B, C, H, W = 1, 512, 54, 96
    x = torch.randn(B, H, W, C, generator=g).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(DEV)
    sc = (torch.rand(C, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(C, generator=g) * 0.2).to(DEV)
    res = torch.randn(B, H, W, C, generator=g).to(DEV) if with_residual else None
    ref = conv3x3_f64(x.permute(0, 3, 1, 2), w) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    ref = ref.clamp_min(0).permute(0, 2, 3, 1)
    tol = 1e-4 * max(1.0, float(ref.abs().max()))
    conv = PackedConv(w, pad=1, scale=sc, shift=sh, relu=True)
    y = conv(x, residual=res)                                               # autotuned, as in the model
    choice = [v for k, v in conv._tile_cache.items()]
    assert float((y.double() - ref).abs().max()) < tol, choice
    for sk in (1, 2, 3, 4, 6, 8):
        y = conv(x, residual=res, tile=TILE_WINO, split_k=sk)
        assert float((y.double() - ref).abs().max()) < tol, sk
    for tile in (1, 2, 3, 4):
        for sk in (1, 4):
            y = conv(x, residual=res, tile=tile, split_k=sk)
            assert float((y.double() - ref).abs().max()) < tol, (tile, sk)


# ------------------------------------------------------------------------------------------------ whole model
CONFS = {"cfg2": S.r50_256_conf, "cfg3": S.r101_512_conf, "cfg5": S.bsm_r101_256_conf}


@pytest.fixture(scope="module", params=list(CONFS))
def full(request):
    """Model + one synthetic frame + the oracle's outputs at the full size of a BASELINE config (CPU: ~5-25 s)."""
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = CONFS[request.param]()
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    # last BatchNorm of every residual block scaled by 0.3 (mmdet zero-initialises it): activations stay O(1-10) through
    # the 100+ layers, so the absolute 1e-3 bar is meaningful (with gamma ~ 1 they reach 1e2 and 1e-3 absolute would be
    # a 1e-5 relative bar, below float32 rounding noise: test_fp32_noise_regime_against_float64 covers that regime)
    S.randomize_norm_stats_(m, 0, residual_gamma=0.3)
    imgs = S.make_images(1, bc['final_dim'], seed=7)
    mats = S.make_mats(1)
    keep = {}
    ref = TM.bevheight_forward(m.state_dict(), bc, hc, imgs, mats, keep)
    keep = {k: keep[k] for k in ('geom_xyz', 'bev')}
    return dict(name=request.param, m=m.to(DEV), imgs=imgs.to(DEV), mats={k: v.to(DEV) for k, v in mats.items()},
                ref=ref, keep=keep, bc=bc, hc=hc)


def _errors(preds, ref):
    worst, scale = 0.0, 0.0
    for t in range(len(ref)):
        for k, v in ref[t][0].items():
            worst = max(worst, float((preds[t][0][k].cpu() - v).abs().max()))
            scale = max(scale, float(v.abs().max()))
    return worst, scale


def test_full_size_fp32_parity(full):
    """north_star: voxel indices bit-exact, BEV features and box regressions within 1e-3 (fp32)."""
    m = full['m']
    with torch.no_grad():
        bev = m.backbone(full['imgs'], full['mats'])
        preds = m(full['imgs'], full['mats'])
        geom, _ = m.backbone.calibration(full['mats'], 0)
    assert np.array_equal(geom.cpu().numpy(), full['keep']['geom_xyz'])
    torch.testing.assert_close(bev.cpu(), full['keep']['bev'], rtol=1e-3, atol=1e-3)
    worst, scale = _errors(preds, full['ref'])
    print(f"{full['name']}: fp32 max |hip - oracle| = {worst:.3e} (max |ref| {scale:.2f})")
    assert worst < 1e-3, (full['name'], worst)


def test_fp32_noise_regime_against_float64():
    """cfg-2 at full size with UNSCALED residual branches (gamma ~ 1, activations ~ 1e2): the HIP outputs and the
    torch-CPU fp32 oracle differ by ~1.5e-3 -- which is the rounding noise of the oracle itself.  Yardstick: the same
    forward in float64 (torch on the GPU).  The HIP path must be as close to it as the fp32 oracle is (within 2x), and
    within 1e-3 relative to the output scale."""
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.r50_256_conf()
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(m, 0)
    imgs, mats = S.make_images(1, bc['final_dim'], seed=7), S.make_mats(1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref32 = TM.bevheight_forward(sd, bc, hc, imgs, mats)
    ref64 = TM.bevheight_forward_highprec(sd, bc, hc, imgs, mats, device=DEV)
    m = m.to(DEV)
    with torch.no_grad():
        preds = m(imgs.to(DEV), {k: v.to(DEV) for k, v in mats.items()})
    e_hip = max(float((preds[t][0][k].double() - ref64[t][0][k]).abs().max()) for t in range(6) for k in ref32[t][0])
    e_cpu = max(float((ref32[t][0][k].double() - ref64[t][0][k].cpu()).abs().max()) for t in range(6) for k in ref32[t][0])
    scale = max(float(ref64[t][0][k].abs().max()) for t in range(6) for k in ref32[t][0])
    print(f"gamma~1 regime: |hip - f64| = {e_hip:.3e}, |torch-cpu fp32 - f64| = {e_cpu:.3e}, output scale {scale:.1f}")
    assert e_hip <= 2.0 * e_cpu + 1e-6 * scale
    assert e_hip <= 1e-3 * scale


def test_checkpoint_like_weights_against_float64():
    """cfg-2 at full size with weights shaped like a trained checkpoint (``synthetic.checkpoint_like_``:
    small random last-BN gammas in every residual block, BatchNorm running_var spread over 1e-2 .. 1e2 with the producing
    convolution's weights scaled along) instead of hand-shaped initial values.  Bars: the HIP outputs are as close to a float64
    evaluation as the torch-CPU fp32 oracle is (within 2x) -- the relative bar that holds for any checkpoint -- AND, since these
    activations stay O(1-10), within the north star's absolute 1e-3 of the fp32 oracle; voxel indices bit-exact."""
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.r50_256_conf()
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.checkpoint_like_(m, 3)
    var = torch.cat([b.running_var.flatten() for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d)])
    assert float(var.min()) < 2e-2 and float(var.max()) > 50.0                     # four decades of channel scales
    imgs, mats = S.make_images(1, bc['final_dim'], seed=11), S.make_mats(1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    keep = {}
    ref32 = TM.bevheight_forward(sd, bc, hc, imgs, mats, keep)
    ref64 = TM.bevheight_forward_highprec(sd, bc, hc, imgs, mats, device=DEV)
    m = m.to(DEV)
    dmats = {k: v.to(DEV) for k, v in mats.items()}
    with torch.no_grad():
        preds = m(imgs.to(DEV), dmats)
        geom, _ = m.backbone.calibration(dmats, 0)
    assert np.array_equal(geom.cpu().numpy(), keep['geom_xyz'])
    e_hip = max(float((preds[t][0][k].double() - ref64[t][0][k]).abs().max()) for t in range(6) for k in ref32[t][0])
    e_cpu = max(float((ref32[t][0][k].double() - ref64[t][0][k].cpu()).abs().max()) for t in range(6) for k in ref32[t][0])
    scale = max(float(ref64[t][0][k].abs().max()) for t in range(6) for k in ref32[t][0])
    worst, _ = _errors(preds, ref32)
    print(f"checkpoint-like weights: |hip - f64| = {e_hip:.3e}, |torch-cpu fp32 - f64| = {e_cpu:.3e}, |hip - oracle| = {worst:.3e}, "
          f"output scale {scale:.2f}")
    assert e_hip <= 2.0 * e_cpu + 1e-6 * scale
    assert worst < 1e-3


# bf16 tolerance.  Operands and (in bf16-activation mode) the tensors between the ResNet layers are rounded to bf16 (relative
# 2^-9 = 2e-3) and accumulated in f32; over the K = 576..4608 terms of a layer the rounding errors average out (relative
# error of a layer output ~ 2^-9, not K x 2^-9) and compound over the ~60-110 convolutions between image and prediction maps
# like a random walk: ~2e-3 x sqrt(110) = 2e-2 of the activation scale is the expected order for the R101 models.  Measured
# at full size (printed below, recorded in DESIGN.md 3.1d): 6e-3 (cfg-2), 5e-3 (cfg-3), 7e-3 (cfg-5) of max |ref|.  The
# bound asserted is 2e-2 of max(1, |ref|max): 3x head-room over the measurements, the order of the estimate.
BF16_TOL = 2e-2


def test_full_size_bf16_mode(full):
    """The bf16-MFMA mode at full size (BASELINE configs[2] / [4] name bf16): same voxel indices (geometry never runs on
    MFMA), predictions within BF16_TOL of the fp32 oracle and not identical to the fp32 path's."""
    m = full['m']
    old = hip_ops.MFMA_BF16
    hip_ops.MFMA_BF16 = True
    m.refresh()
    try:
        with torch.no_grad():
            preds = m(full['imgs'], full['mats'])
            geom, _ = m.backbone.calibration(full['mats'], 0)
        torch.cuda.synchronize()
    finally:
        hip_ops.MFMA_BF16 = old
        m.refresh()
    assert np.array_equal(geom.cpu().numpy(), full['keep']['geom_xyz'])
    worst, scale = _errors(preds, full['ref'])
    print(f"{full['name']}: bf16 max |hip - oracle| = {worst:.3e} (max |ref| {scale:.2f})")
    assert 1e-5 < worst < BF16_TOL * max(1.0, scale), (full['name'], worst, scale)


# ------------------------------------------------------------------------------------------------ BASELINE batch sizes
# BASELINE configs[2] is batch 4 per GPU (and configs[3] / [4] shard 4 frames per rank): batch 4 changes M of every
# convolution -- hence every measured (algorithm, tile, split-K) choice and grid size -- and the 4x voxel plan / lifted
# tensor.  Four DIFFERENT frames and calibrations, each compared with its own oracle forward.
@pytest.fixture(scope="module", params=["cfg3", "cfg5"])
def full_b4(request):
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = CONFS[request.param]()
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(m, 0, residual_gamma=0.3)
    imgs = S.make_images(4, bc['final_dim'], seed=11)
    mats = S.make_mats(4)                                      # vary=True: four different camera poses
    assert not torch.equal(mats['sensor2ego_mats'][0], mats['sensor2ego_mats'][3])
    keep = {}
    ref = TM.bevheight_forward(m.state_dict(), bc, hc, imgs, mats, keep)
    return dict(name=request.param, m=m.to(DEV), imgs=imgs.to(DEV), mats={k: v.to(DEV) for k, v in mats.items()},
                ref=ref, geom=keep['geom_xyz'], bc=bc, hc=hc)


def _per_sample_errors(preds, ref, batch):
    """(max |err|, max |ref|) per sample over all prediction maps."""
    out = []
    for b in range(batch):
        worst, scale = 0.0, 0.0
        for t in range(len(ref)):
            for k, v in ref[t][0].items():
                worst = max(worst, float((preds[t][0][k][b].float().cpu() - v[b]).abs().max()))
                scale = max(scale, float(v[b].abs().max()))
        out.append((worst, scale))
    return out


def test_batch4_fp32_parity(full_b4):
    """cfg-3 / cfg-5 at batch 4, fp32: voxel indices bit-exact for all four calibrations, every frame within 1e-3."""
    m = full_b4['m']
    with torch.no_grad():
        preds = m(full_b4['imgs'], full_b4['mats'])
        geom, _ = m.backbone.calibration(full_b4['mats'], 0)
    assert np.array_equal(geom.cpu().numpy(), full_b4['geom'])
    errs = _per_sample_errors(preds, full_b4['ref'], 4)
    print(f"{full_b4['name']} batch 4 fp32: per-frame max |hip - oracle| = {[f'{e:.2e}' for e, _ in errs]}")
    assert all(e < 1e-3 for e, _ in errs), (full_b4['name'], errs)
    # the frames are really different (a batch that silently repeated frame 0 would pass the per-frame check on frame 0 only)
    assert float((preds[0][0]['heatmap'][0] - preds[0][0]['heatmap'][3]).abs().max()) > 1e-3


def test_batch4_bf16_mode(full_b4):
    """cfg-3 / cfg-5 at batch 4 in the bf16-MFMA mode (the dtype BASELINE configs[2] / [4] name; the launches
    bench.py's other_configs records time): every frame within BF16_TOL of its fp32 oracle forward."""
    m = full_b4['m']
    old = hip_ops.MFMA_BF16
    hip_ops.MFMA_BF16 = True
    m.refresh()
    try:
        with torch.no_grad():
            preds = m(full_b4['imgs'], full_b4['mats'])
            geom, _ = m.backbone.calibration(full_b4['mats'], 0)
        torch.cuda.synchronize()
    finally:
        hip_ops.MFMA_BF16 = old
        m.refresh()
    assert np.array_equal(geom.cpu().numpy(), full_b4['geom'])
    errs = _per_sample_errors(preds, full_b4['ref'], 4)
    print(f"{full_b4['name']} batch 4 bf16: per-frame max |hip - oracle| / max |ref| = {[f'{e / max(1.0, s):.2e}' for e, s in errs]}")
    assert all(1e-5 < e < BF16_TOL * max(1.0, s) for e, s in errs), (full_b4['name'], errs)
