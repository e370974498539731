"""The device path of the calibration cache (sgv3d_amd/calibration.py, round 6): calibration tensors that are NEW OBJECTS with the
numbers of the last frame -- what the reference harness hands over per batch for a static camera
(exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:244-247) -- are recognised by one compare launch on the device;
the geometry kernels and the gate MLPs take its flag and skip themselves.  No host synchronisation, results bitwise those of the
recomputing path, a changed calibration recomputes."""
import pytest
import torch

from sgv3d_amd import hip_ops
from sgv3d_amd import synthetic as S
from sgv3d_amd.calibration import CalibrationCache
from sgv3d_amd.layers.backbones import lss_fpn
from sgv3d_amd.pipeline import eager_forward

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _model(bsm=False):
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.small_bsm_conf(depth=18) if bsm else S.small_conf(depth=18)
    torch.manual_seed(0)
    m = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(m, 1)
    return m.to(DEV), bc


def test_gated_kernels_skip_on_a_zero_flag_and_run_on_a_one():
    g = torch.Generator().manual_seed(0)
    x, w, b = (torch.randn(s, generator=g).to(DEV) for s in ((3, 40), (24, 40), (24,)))
    want = hip_ops.dense(x, w, None, b, hip_ops.ACT_RELU)
    out = torch.full((3, 24), 7.0, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip_ops.dense(x, w, None, b, hip_ops.ACT_RELU, out=out, run=flag)
    assert float(out.min()) == 7.0 == float(out.max())
    flag.fill_(1)
    hip_ops.dense(x, w, None, b, hip_ops.ACT_RELU, out=out, run=flag)
    assert torch.equal(out, want)
    m, bc = _model()
    mats = {k: v.to(DEV) for k, v in S.make_mats(2, scale=bc['final_dim'][0] / 864).items()}
    args = (mats['sensor2ego_mats'][:, 0], mats['sensor2virtual_mats'][:, 0], mats['intrin_mats'][:, 0], mats['ida_mats'][:, 0],
            mats['reference_heights'][:, 0], mats['bda_mat'])
    want = m.backbone.get_geometry_voxel_index(*args)
    buf = torch.full_like(want, -77)
    flag.fill_(0)
    m.backbone.get_geometry_voxel_index(*args, out=buf, run=flag)
    assert int(buf.min()) == -77 == int(buf.max())
    flag.fill_(1)
    m.backbone.get_geometry_voxel_index(*args, out=buf, run=flag)
    assert torch.equal(buf, want)


def test_compare_kernel_sees_equal_changed_and_forced():
    m, bc = _model()
    cc = CalibrationCache().entry(0)
    mats = {k: v.to(DEV) for k, v in S.make_mats(2, scale=bc['final_dim'][0] / 864).items()}
    names = ('sensor2ego_mats', 'sensor2virtual_mats', 'intrin_mats', 'ida_mats', 'reference_heights', 'bda_mat')
    srcs = [mats[k] for k in names]
    f = m.backbone._calibration_changed(cc, srcs, force=False)
    assert int(f) == 1                                           # first sight
    assert int(m.backbone._calibration_changed(cc, [t.clone() for t in srcs], force=False)) == 0
    assert int(m.backbone._calibration_changed(cc, [t.clone() for t in srcs], force=True)) == 1
    moved = [t.clone() for t in srcs]
    moved[4][0, 0, 0] += 1e-3                                    # one reference height, one ulp-scale edit elsewhere
    assert int(m.backbone._calibration_changed(cc, moved, force=False)) == 1
    assert int(m.backbone._calibration_changed(cc, [t.clone() for t in moved], force=False)) == 0     # the copy followed
    assert int(m.backbone._calibration_changed(cc, srcs, force=False)) == 1
    nan = [t.clone() for t in srcs]
    nan[0][0, 0, 0, 0, 0] = float('nan')                         # bit compare: NaN == NaN, -0.0 != 0.0
    assert int(m.backbone._calibration_changed(cc, nan, force=False)) == 1
    assert int(m.backbone._calibration_changed(cc, [t.clone() for t in nan], force=False)) == 0
    # tensors it cannot compare word-wise: no flag, the callers recompute
    assert m.backbone._calibration_changed(cc, [srcs[0].transpose(-1, -2)] + srcs[1:], force=False) is None


@pytest.mark.parametrize("bsm", [False, True])
def test_fresh_tensors_with_the_same_numbers_skip_the_refresh_and_change_nothing(bsm):
    m, bc = _model(bsm)
    imgs = S.make_images(2, bc['final_dim'], seed=3).to(DEV)
    mats = {k: v.to(DEV) for k, v in S.make_mats(2, scale=bc['final_dim'][0] / 864).items()}
    key = lambda out: torch.cat([v.reshape(-1) for t in out for v in t[0].values()])
    old = m.backbone.calib_cache
    m.backbone.calib_cache = cache = CalibrationCache()
    try:
        with torch.no_grad(), eager_forward(m):
            a = key(m(imgs, mats))
            cc = cache.entry(0)
            builds = cc.plan.builds()
            # the refresh for new objects with the same numbers launches, and skips itself: poisoned buffers stay poisoned
            geom_ok, gates_ok = cc.geom.clone(), [g.clone() for g in cc.gates]
            cc.geom.fill_(-5)
            for g in cc.gates:
                g.fill_(9.0)
            m.backbone.calibration({k: v.clone() for k, v in mats.items()}, 0)
            assert int(cc.geom.min()) == -5 == int(cc.geom.max()) and all(float(g.min()) == 9.0 for g in cc.gates)
            assert int(cc.changed) == 0
            # ... while a different calibration recomputes them (one reference height moved)
            moved = {k: v.clone() for k, v in mats.items()}
            moved['reference_heights'] += 0.25
            m.backbone.calibration(moved, 0)
            assert int(cc.changed) == 1 and int(cc.geom.min()) != -5 and all(float(g.max()) <= 1.0 for g in cc.gates)
            # and the original one again, through new objects: recomputed (the numbers differ from the last ones), the original state
            m.backbone.calibration({k: v.clone() for k, v in mats.items()}, 0)
            assert torch.equal(cc.geom, geom_ok) and all(torch.equal(x, y) for x, y in zip(cc.gates, gates_ok))
            # whole forwards: new objects / same numbers == the first forward bitwise, without a plan rebuild
            b = key(m(imgs, {k: v.clone() for k, v in mats.items()}))
            c = key(m(imgs, {k: v.clone() for k, v in mats.items()}))
            assert torch.equal(a, b) and torch.equal(a, c)
            d = key(m(imgs, moved))
            lss_fpn.GATED_CALIBRATION = False
            try:
                m.backbone.calib_cache = CalibrationCache()
                d_cold = key(m(imgs, moved))
                a_cold = key(m(imgs, {k: v.clone() for k, v in mats.items()}))
            finally:
                lss_fpn.GATED_CALIBRATION = True
                m.backbone.calib_cache = cache
            assert torch.equal(d, d_cold) and torch.equal(a, a_cold) and not torch.equal(a, d)
            assert builds >= 1
    finally:
        m.backbone.calib_cache = old
