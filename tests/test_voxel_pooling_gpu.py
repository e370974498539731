"""GPU: HIP voxel pooling (C ABI + python op mirror) against the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import geometry_ref as G
from oracle import voxel_pooling_ref as VPO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VP_CASES = ["tiny", "b2_c80", "z2", "dups", "all_out"]


@pytest.fixture(params=["rule", "slot", "vox"])
def gather_kernel(request, hip):
    """Both gather kernels on the same data: the slot-balanced one of round 3 and the voxel-owner one of round 4 (the default
    picks per form and grid density, csrc/voxel_pooling.hip::launch_gather)."""
    lib = hip.load()
    hip.check(lib.sgv3d_voxel_pooling_select_kernel({"rule": 0, "slot": 1, "vox": 2}[request.param]), "select")
    yield request.param
    hip.check(lib.sgv3d_voxel_pooling_select_kernel(0), "select")


def _run_abi(hip, geom, feats, voxel_num, mode, sort=True):
    """Call straight through the C ABI. Returns NCHW out (numpy), pos_memo (numpy)."""
    lib = hip.load()
    B = geom.shape[0]
    C = feats.shape[-1]
    g = torch.from_numpy(np.ascontiguousarray(geom, np.int32).reshape(B, -1, 3)).to(DEV)
    f = torch.from_numpy(np.ascontiguousarray(feats, np.float32).reshape(B, -1, C)).to(DEV)
    N = g.shape[1]
    X, Y, Z = (int(v) for v in voxel_num)
    pm = torch.full((B, N, 3), -1, dtype=torch.int32, device=DEV)
    st = hip.stream_handle()
    if mode == "atomic":
        out = torch.zeros(B, Y, X, C, device=DEV)
        hip.check(lib.sgv3d_voxel_pooling_forward_atomic(B, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(),
                                                         pm.data_ptr(), st), "atomic")
    elif mode == "level1":
        # the drop-in symbol the reference's wrapper reaches: the first call builds the library-owned plan, the second
        # one runs the steady-state form (device-side compare + gated gather); both must give the same answer
        first = torch.zeros(B, Y, X, C, device=DEV)
        pm0 = torch.full((B, N, 3), -1, dtype=torch.int32, device=DEV)
        hip.check(lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), first.data_ptr(),
                                                  pm0.data_ptr(), st), "level1 (first call)")
        out = torch.zeros(B, Y, X, C, device=DEV)
        hip.check(lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(),
                                                  pm.data_ptr(), st), "level1 (steady state)")
        torch.cuda.synchronize()
        assert torch.equal(first, out) and torch.equal(pm0, pm)
    else:
        nbytes = lib.sgv3d_voxel_plan_bytes(B, N, X, Y)
        plan = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        out = torch.full((B, Y, X, C), float("nan"), device=DEV)   # must be fully overwritten
        hip.check(lib.sgv3d_voxel_plan_build(B, N, X, Y, Z, g.data_ptr(), pm.data_ptr(), plan.data_ptr(), nbytes,
                                             1 if sort else 0, st), "plan")
        nws = lib.sgv3d_voxel_pooling_workspace_bytes(B, N, C)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        hip.check(lib.sgv3d_voxel_pooling_forward_planned(B, N, C, X, Y, plan.data_ptr(), f.data_ptr(),
                                                          out.data_ptr(), ws.data_ptr(), nws, st), "planned")
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).contiguous().cpu().numpy(), pm.cpu().numpy()


@pytest.mark.parametrize("name", VP_CASES)
@pytest.mark.parametrize("mode", ["atomic", "planned", "level1"])
def test_golden_exact(hip, golden, name, mode):
    vp = golden["voxel_pooling"]
    out, pm = _run_abi(hip, vp[f"{name}/geom_xyz"], vp[f"{name}/feats"], vp[f"{name}/voxel_num"], mode)
    assert np.array_equal(out, vp[f"{name}/out"])            # integer-valued floats: bit-exact
    _, pm_ref = VPO.forward(vp[f"{name}/geom_xyz"], vp[f"{name}/feats"], vp[f"{name}/voxel_num"])
    assert np.array_equal(pm, pm_ref)


@pytest.mark.parametrize("mode", ["atomic", "planned"])
def test_golden_randn_tolerance(hip, golden, mode):
    vp = golden["voxel_pooling"]
    out, _ = _run_abi(hip, vp["b2_c80_randn/geom_xyz"], vp["b2_c80_randn/feats"], vp["b2_c80_randn/voxel_num"], mode)
    np.testing.assert_allclose(out, vp["b2_c80_randn/out"], rtol=1e-3, atol=1e-3)   # north_star: 1e-3 fp32


@pytest.mark.parametrize("C", [80, 87, 4, 1, 260])
@pytest.mark.parametrize("mode", ["atomic", "planned", "level1"])
def test_random_vs_oracle_exact(hip, C, mode, gather_kernel):
    """Ragged / colliding / out-of-range indices, channel counts incl. C%4!=0 and rows wider than a wave."""
    rng = np.random.default_rng(C)
    B, N, X, Y, Z = 2, 3001, 13, 9, 2
    geom = rng.integers(-3, 16, size=(B, N, 3)).astype(np.int32)
    geom[0, :50] = [5, 5, 0]                                   # a 50-fold collision
    feats = rng.integers(-8, 9, size=(B, N, C)).astype(np.float32)
    out, pm = _run_abi(hip, geom, feats, (X, Y, Z), mode)
    ref, pm_ref = VPO.forward(geom, feats, (X, Y, Z))
    assert np.array_equal(out, ref) and np.array_equal(pm, pm_ref)


def test_long_segments_sorted(hip, gather_kernel):
    """Segments of 1, 64, 65, 300 and 9000 points: the plan's per-voxel lists come out ascending
    (covers the in-register-size, LDS and global-memory bitonic paths)."""
    lib = hip.load()
    lens = [1, 64, 65, 300, 9000, 2, 3, 257]
    X, Y, Z = len(lens), 1, 1
    vox = np.concatenate([np.full(n, i) for i, n in enumerate(lens)])
    rng = np.random.default_rng(0)
    rng.shuffle(vox)
    N = len(vox)
    geom = np.zeros((1, N, 3), np.int32)
    geom[0, :, 0] = vox
    g = torch.from_numpy(geom).to(DEV)
    nbytes = lib.sgv3d_voxel_plan_bytes(1, N, X, Y)
    plan = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    hip.check(lib.sgv3d_voxel_plan_build(1, N, X, Y, Z, g.data_ptr(), None, plan.data_ptr(), nbytes, 1,
                                         hip.stream_handle()), "plan")
    torch.cuda.synchronize()
    ints = plan.cpu().numpy().view(np.int32)
    V = X * Y
    al = lambda b: (b + 255) & ~255
    off_cur = al(4 * (V + 1))
    off_order = al(off_cur + 4 * (V + 1))
    seg = ints[:V + 1]
    order = ints[off_order // 4: off_order // 4 + N]
    assert list(np.diff(seg)) == lens
    for v in range(V):
        s = order[seg[v]:seg[v + 1]]
        assert np.array_equal(s, np.nonzero(vox == v)[0]), f"segment {v} (len {lens[v]}) not ascending"


def test_planned_is_bitwise_reproducible(hip):
    rng = np.random.default_rng(5)
    B, N, C, X, Y = 1, 20000, 80, 16, 16
    geom = rng.integers(-1, 17, size=(B, N, 3)).astype(np.int32)
    geom[..., 2] = 0
    feats = rng.standard_normal((B, N, C)).astype(np.float32)
    a, _ = _run_abi(hip, geom, feats, (X, Y, 1), "planned")
    b, _ = _run_abi(hip, geom, feats, (X, Y, 1), "planned")
    assert np.array_equal(a, b)
    ref, _ = VPO.forward(geom, feats, (X, Y, 1))
    np.testing.assert_allclose(a, ref, rtol=1e-3, atol=1e-3)


def _cfg2_geom(golden):
    geo = golden["geometry"]
    n = "dair_p11_h5.5"
    vs, vc, vn = G.voxel_params([0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
    fr = G.create_frustum((864, 1536), 16, [-2.0, 0.0, 90])
    gi, _ = G.geom_xyz_for_camera(fr, geo[f"{n}/sensor2ego"], geo[f"{n}/sensor2virtual"], geo[f"{n}/intrin"],
                                  geo[f"{n}/ida"], geo[f"{n}/reference_height"], geo[f"{n}/bda"], vc, vs)
    return gi[None], vn


@pytest.mark.parametrize("mode", ["atomic", "planned", "level1"])
def test_cfg2_full_size_properties(hip, golden, mode):
    """BASELINE cfg-2 size (N=466 560, C=80, 256x256): exact vs the oracle on integer-valued features,
    plus size-independent properties: linearity and conservation of mass."""
    geom, vn = _cfg2_geom(golden)
    rng = np.random.default_rng(1)
    N = geom.shape[1] * geom.shape[2] * geom.shape[3]
    f1 = rng.integers(-4, 5, size=(1, N, 80)).astype(np.float32)
    f2 = rng.integers(-4, 5, size=(1, N, 80)).astype(np.float32)
    o1, pm = _run_abi(hip, geom, f1, vn, mode)
    ref, pm_ref = VPO.forward(geom, f1, vn)
    assert np.array_equal(o1, ref) and np.array_equal(pm, pm_ref)
    o2, _ = _run_abi(hip, geom, f2, vn, mode)
    o12, _ = _run_abi(hip, geom, f1 + 2 * f2, vn, mode)
    assert np.array_equal(o12, o1 + 2 * o2)                                   # linearity
    kept = pm.reshape(-1, 3)[:, 0] != -1
    assert np.array_equal(o1.sum(axis=(0, 2, 3)), f1.reshape(-1, 80)[kept].sum(0))  # mass conservation
    assert abs(kept.mean() - 0.7432) < 1e-3


@pytest.mark.parametrize("mode", ["atomic", "planned"])
def test_python_op_forward_backward(hip, golden, mode):
    """The reference-shaped operator: permuted-view output, (None, grad, None) backward."""
    from sgv3d_amd.ops.voxel_pooling import voxel_pooling, set_mode
    vp = golden["voxel_pooling"]
    set_mode(mode)
    try:
        for name in ["b2_c80", "z2", "dups", "all_out"]:
            geom = torch.from_numpy(vp[f"{name}/geom_xyz"]).to(DEV)
            feats = torch.from_numpy(vp[f"{name}/feats"]).to(DEV).requires_grad_(True)
            vnum = torch.from_numpy(vp[f"{name}/voxel_num"]).to(DEV)       # CUDA LongTensor like lss_fpn.py:491
            out = voxel_pooling(geom, feats, vnum)
            assert tuple(out.shape) == vp[f"{name}/out"].shape
            assert np.array_equal(out.detach().cpu().numpy(), vp[f"{name}/out"])
            out.backward(torch.from_numpy(vp[f"{name}/grad_out"]).to(DEV))
            assert np.array_equal(feats.grad.cpu().numpy(), vp[f"{name}/grad_feats"])
            with torch.no_grad():
                out2 = voxel_pooling(geom, feats.detach(), vnum)
            assert torch.equal(out2, out.detach())
    finally:
        set_mode("planned")


def test_python_op_errors(hip):
    from sgv3d_amd.ops.voxel_pooling import voxel_pooling
    g = torch.zeros(1, 4, 3, dtype=torch.int32, device=DEV)
    f = torch.zeros(1, 4, 8, device=DEV)
    with pytest.raises(RuntimeError):
        voxel_pooling(g.cpu(), f, (2, 2, 1))                   # not a CUDA tensor
    with pytest.raises(AssertionError):
        voxel_pooling(g, f.transpose(1, 2).contiguous().transpose(1, 2), (2, 2, 1))   # non-contiguous
    with pytest.raises(RuntimeError):
        voxel_pooling(g.long(), f, (2, 2, 1))                  # wrong dtype


def test_ext_shim_runs_reference_wrapper_protocol(hip, golden):
    """voxel_pooling_ext.voxel_pooling_forward_wrapper: same 10-argument call the reference makes."""
    from sgv3d_amd.ops.voxel_pooling import voxel_pooling_ext as ext
    vp = golden["voxel_pooling"]
    name = "b2_c80"
    geom = torch.from_numpy(vp[f"{name}/geom_xyz"]).to(DEV).reshape(2, -1, 3)
    feats = torch.from_numpy(vp[f"{name}/feats"]).to(DEV).reshape(2, -1, 80)
    X, Y, Z = (int(v) for v in vp[f"{name}/voxel_num"])
    out = feats.new_zeros(2, Y, X, 80)
    pm = geom.new_ones(2, geom.shape[1], 3) * -1
    assert ext.voxel_pooling_forward_wrapper(2, geom.shape[1], 80, X, Y, Z, geom, feats, out, pm) == 1
    assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), vp[f"{name}/out"])
    with pytest.raises(RuntimeError):
        ext.voxel_pooling_forward_wrapper(2, geom.shape[1], 80, X, Y, Z, geom.cpu(), feats, out, pm)


def test_compiled_pybind_ext_runs_reference_wrapper_protocol(hip, golden):
    """The COMPILED pybind11 module (src/voxel_pooling_ext.cpp): the 10-argument call of the reference's
    voxel_pooling.py:41-52 on the reference-generated goldens, on the current (non-default) stream; error behaviour of
    its CHECK_INPUT macros."""
    from sgv3d_amd.ops.voxel_pooling import compiled_ext
    ext = compiled_ext.load()
    vp = golden["voxel_pooling"]
    for name in ("tiny", "b2_c80", "z2", "dups", "all_out"):
        geom = torch.from_numpy(vp[f"{name}/geom_xyz"]).to(DEV)
        B, C = geom.shape[0], vp[f"{name}/feats"].shape[-1]
        geom = geom.reshape(B, -1, 3)
        feats = torch.from_numpy(vp[f"{name}/feats"]).to(DEV).reshape(B, -1, C)
        X, Y, Z = (int(v) for v in vp[f"{name}/voxel_num"])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out = feats.new_zeros(B, Y, X, C)
            pm = geom.new_full((B, geom.shape[1], 3), -1)
            assert ext.voxel_pooling_forward_wrapper(B, geom.shape[1], C, X, Y, Z, geom, feats, out, pm) == 1
        s.synchronize()
        assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), vp[f"{name}/out"]), name
    with pytest.raises(RuntimeError, match="CUDAtensor"):
        ext.voxel_pooling_forward_wrapper(B, geom.shape[1], C, X, Y, Z, geom.cpu(), feats, out, pm)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.voxel_pooling_forward_wrapper(B, geom.shape[1], C, X, Y, Z, geom, feats.transpose(1, 2), out, pm)
    with pytest.raises(RuntimeError):                                   # dtype mismatch: data_ptr<float>() on an int tensor
        ext.voxel_pooling_forward_wrapper(B, geom.shape[1], C, X, Y, Z, geom, feats.int(), out, pm)


def test_fused_lift_splat_matches_materialised(hip, golden, gather_kernel):
    from sgv3d_amd.ops.voxel_pooling import VoxelPlan
    rng = np.random.default_rng(3)
    B, D, P, C, X, Y = 2, 6, 35, 80, 8, 7
    geom = torch.from_numpy(rng.integers(-2, 10, size=(B, D * P, 3)).astype(np.int32)).to(DEV)
    geom[..., 2] = 0
    prob = torch.from_numpy(rng.integers(0, 4, size=(B, D, P)).astype(np.float32)).to(DEV)
    ctx = torch.from_numpy(rng.integers(-4, 5, size=(B, P, C)).astype(np.float32)).to(DEV)
    lifted = (prob[..., None] * ctx[:, None]).reshape(B, D * P, C).contiguous()
    plan = VoxelPlan(geom, (X, Y, 1))
    assert torch.equal(plan.lift_splat(prob, ctx), plan.pool(lifted))


@pytest.mark.parametrize("C,ld", [(80, 96), (88, 96), (24, 32)])
def test_fused_lift_splat_random_data_and_bf16_forms(hip, C, ld, gather_kernel):
    """The fused gather on random (non-integer) data: f32 bitwise the lift + pool result (products rounded before the add, same
    order); the bf16 hand-off output = the f32 sums rounded once, padding channels zero; bf16 context rows = the f32 kernel on the
    rounded context."""
    from sgv3d_amd.ops.voxel_pooling import VoxelPlan
    rng = np.random.default_rng(C)
    B, D, P, X, Y = 2, 9, 77, 10, 9
    geom = torch.from_numpy(rng.integers(-2, 12, size=(B, D * P, 3)).astype(np.int32)).to(DEV)
    geom[..., 2] = 0
    geom[0, : D * P // 3, :2] = 3                                   # one long run (more slots than a wave's window)
    prob = torch.from_numpy(rng.random(size=(B, D, P)).astype(np.float32)).to(DEV)
    ctx = torch.from_numpy(rng.standard_normal(size=(B, P, C)).astype(np.float32)).to(DEV)
    lifted = (prob[..., None] * ctx[:, None]).reshape(B, D * P, C).contiguous()
    plan = VoxelPlan(geom, (X, Y, 1))
    want = plan.pool(lifted)
    got = plan.lift_splat(prob, ctx)
    assert torch.equal(got, want)
    ob = plan.lift_splat(prob, ctx, out_bf16_ld=ld)
    assert ob.dtype == torch.bfloat16 and tuple(ob.shape) == (B, Y, X, ld)
    assert torch.equal(ob[..., :C], want.bfloat16()) and float(ob[..., C:].float().abs().max()) == 0
    cb = ctx.bfloat16()
    ob2 = plan.lift_splat(prob, cb, out_bf16_ld=ld)
    want2 = plan.lift_splat(prob, cb.float())
    assert torch.equal(ob2[..., :C], want2.bfloat16()) and float(ob2[..., C:].float().abs().max()) == 0


def test_cached_plan_rebuilds_only_on_change(hip):
    """VoxelPlan(cached=True): the device-side compare skips the build while geom_xyz is bytewise unchanged (also for a
    fresh tensor object with the same content), rebuilds after a single changed index, and every result equals the
    uncached plan's bit for bit.  The operator-level call keeps one such plan per (stream, sizes)."""
    from sgv3d_amd.ops.voxel_pooling import VoxelPlan, voxel_pooling
    import sys
    VPM = sys.modules['sgv3d_amd.ops.voxel_pooling.voxel_pooling']     # (the package rebinds the name to the function)
    rng = np.random.default_rng(77)
    B, N, C, X, Y = 2, 30001, 80, 40, 33           # N*3 not a multiple of 4: exercises the compare tail
    geom = torch.from_numpy(rng.integers(-2, 42, size=(B, N, 3)).astype(np.int32)).to(DEV)
    geom[..., 2] = 0
    feats = torch.from_numpy(rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)).to(DEV)
    plan = VoxelPlan(geom, (X, Y, 1), cached=True)
    want = VoxelPlan(geom, (X, Y, 1)).pool(feats)
    assert plan.builds() == 1 and torch.equal(plan.pool(feats), want)
    plan.rebuild(geom)
    plan.rebuild(geom.clone())
    assert plan.builds() == 1 and torch.equal(plan.pool(feats), want)
    g2 = geom.clone()
    g2[1, N - 1, 0] = (g2[1, N - 1, 0] + 1) % X                                   # the very last point moves one cell
    plan.rebuild(g2)
    assert plan.builds() == 2
    assert torch.equal(plan.pool(feats), VoxelPlan(g2, (X, Y, 1)).pool(feats))
    plan.rebuild(geom)
    assert plan.builds() == 3 and torch.equal(plan.pool(feats), want)
    with pytest.raises(AssertionError):
        VoxelPlan(geom, (X, Y, 1), cached=True, pos_memo=torch.empty(B, N, 3, dtype=torch.int32, device=DEV))
    # operator level: consecutive frames of one camera.  C = 80 goes through the library's own plan cache (the level-1
    # entry): three calls, all through a plan entry, the changed geometry served at once (gated scatter) ...
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")
    s0 = _l1_stats(hip)
    o1 = voxel_pooling(geom, feats, (X, Y, 1))
    o2 = voxel_pooling(geom.clone(), feats, (X, Y, 1))
    o3 = voxel_pooling(g2, feats, (X, Y, 1))
    torch.cuda.synchronize()
    o4 = voxel_pooling(g2, feats, (X, Y, 1))                   # ... and by the rebuilt plan a call later
    s1 = _l1_stats(hip)
    assert s1[0] - s0[0] == 4 and s1[1] - s0[1] == 4 and s1[3] - s0[3] == 2
    assert torch.equal(o1, want.permute(0, 3, 1, 2)) and torch.equal(o2, o1) and not torch.equal(o3, o1) and torch.equal(o4, o3)
    assert torch.equal(o3, VoxelPlan(g2, (X, Y, 1)).pool(feats).permute(0, 3, 1, 2))
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")
    # ... a channel count the level-1 gather does not cover keeps the operator's own cached plan
    VPM._PLAN_CACHE.clear()
    f6 = feats[..., :6].contiguous()
    p1 = voxel_pooling(geom, f6, (X, Y, 1))
    p2 = voxel_pooling(geom.clone(), f6, (X, Y, 1))
    p3 = voxel_pooling(g2, f6, (X, Y, 1))
    (cached_plan,) = VPM._PLAN_CACHE.values()
    assert cached_plan.builds() == 2
    assert torch.equal(p1, want[..., :6].permute(0, 3, 1, 2)) and torch.equal(p2, p1) and not torch.equal(p3, p1)
    VPM._PLAN_CACHE.clear()


@pytest.mark.parametrize("seed", range(12))
def test_random_geometry_fuzz_vs_oracle(hip, seed, gather_kernel):
    """Random sizes / channel counts / grids and point clouds from benign to pathological (everything in one
    voxel, everything out of range, long runs of equal voxels, interleaved duplicates): both kernels exact on
    integer-valued features, pos_memo exact, planned output fully written."""
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 4))
    N = int(rng.choice([1, 7, 64, 500, 4097, 30000]))
    C = int(rng.choice([1, 3, 4, 20, 80, 87, 132]))
    X, Y, Z = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 3))
    kind = seed % 6
    if kind == 0:      # uniform with a halo of out-of-range points
        geom = rng.integers(-3, max(X, Y) + 3, size=(B, N, 3))
        geom[..., 2] = rng.integers(-1, Z + 1, size=(B, N))
    elif kind == 1:    # everything in one voxel
        geom = np.zeros((B, N, 3), np.int64)
        geom[..., 0], geom[..., 1] = X - 1, Y // 2
    elif kind == 2:    # everything out of range
        geom = np.full((B, N, 3), -5, np.int64)
    elif kind == 3:    # long runs of equal voxels (frustum rows)
        base = rng.integers(0, X * Y, size=(B, max(1, N // 37) + 1))
        v = np.repeat(base, 37, axis=1)[:, :N]
        geom = np.stack([v % X, v // X, np.zeros_like(v)], -1)
    elif kind == 4:    # two hot voxels interleaved with noise
        geom = rng.integers(0, max(X, Y), size=(B, N, 3))
        geom[..., 2] = 0
        geom[:, ::2, 0], geom[:, ::2, 1] = 0, 0
        geom[:, 1::4, 0], geom[:, 1::4, 1] = X - 1, Y - 1
    else:              # sorted by voxel (best case for the run aggregation)
        v = np.sort(rng.integers(0, X * Y, size=(B, N)), axis=1)
        geom = np.stack([v % X, v // X, np.zeros_like(v)], -1)
    geom = geom.astype(np.int32)
    feats = rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)
    ref_out, ref_pm = VPO.forward(geom, feats, (X, Y, Z))
    for mode in ("atomic", "planned", "level1"):
        out, pm = _run_abi(hip, geom, feats, (X, Y, Z), mode)
        assert np.array_equal(out, ref_out), (seed, mode, B, N, C, X, Y, Z)
        assert np.array_equal(pm.reshape(ref_pm.shape), ref_pm), (seed, mode)


# ------------------------------------------------------------------------------------------------ owner-computes gather
@pytest.mark.parametrize("C", [80, 64, 24, 88, 256])
def test_gather_runs_cut_by_chunk_and_wave_borders(hip, C, gather_kernel):
    """vp_gather3_kernel: voxel populations chosen around every border of its decomposition -- one chunk (<= 16 slots), one
    wave range (groups x ch slots: 48 at C=80, 56 at C=64, 40 at C=24, 32 at C=88, 16 at C=256), kLongRun = 256 (longer runs
    go to the long-run workgroups), several ranges, and thousands of points -- in shuffled point order, two samples, empty
    voxels between them.  Integer-valued features: exact against the oracle; output fully written (NaN-prefilled)."""
    lens = [1, 2, 15, 16, 17, 31, 32, 33, 39, 40, 41, 47, 48, 49, 55, 56, 57, 95, 96, 97, 255, 256, 257, 258, 300, 511, 513,
            1000, 4999, 1, 1, 3, 64, 128]
    X, Y, Z = len(lens) + 7, 3, 1
    rng = np.random.default_rng(C)
    geoms = []
    for b in range(2):
        cells = rng.permutation(X * Y)[:len(lens)]                       # which voxel gets which population
        vox = np.concatenate([np.full(n, c) for c, n in zip(cells, lens)] + [np.full(200, -1)])
        rng.shuffle(vox)
        gx = np.where(vox >= 0, vox % X, -3)
        geoms.append(np.stack([gx, np.where(vox >= 0, vox // X, 0), np.zeros_like(vox)], -1))
    geom = np.stack(geoms).astype(np.int32)
    feats = rng.integers(-4, 5, size=(2, geom.shape[1], C)).astype(np.float32)
    ref, pm_ref = VPO.forward(geom, feats, (X, Y, Z))
    for mode in ("planned", "level1"):
        out, pm = _run_abi(hip, geom, feats, (X, Y, Z), mode)
        assert np.array_equal(out, ref), (C, mode)
        assert np.array_equal(pm.reshape(pm_ref.shape), pm_ref)


def test_gather_variants_agree_on_cut_runs(hip, gather_kernel):
    """The fused lift-splat form, the bf16-feature form and the bf16-output form of the gather on the same cut / long runs."""
    from sgv3d_amd.ops.voxel_pooling import VoxelPlan
    rng = np.random.default_rng(9)
    B, D, P, C, X, Y = 1, 40, 700, 80, 9, 5
    v = rng.choice(X * Y, size=(B, D * P), p=np.r_[[0.45, 0.2, 0.1], np.full(X * Y - 3, 0.25 / (X * Y - 3))])
    geom = torch.from_numpy(np.stack([v % X, v // X, np.zeros_like(v)], -1).astype(np.int32)).to(DEV)
    prob = torch.from_numpy(rng.integers(0, 3, size=(B, D, P)).astype(np.float32)).to(DEV)
    ctx = torch.from_numpy(rng.integers(-4, 5, size=(B, P, C)).astype(np.float32)).to(DEV)
    lifted = (prob[..., None] * ctx[:, None]).reshape(B, D * P, C).contiguous()
    plan = VoxelPlan(geom, (X, Y, 1))
    want = plan.pool(lifted)
    ref, _ = VPO.forward(geom.cpu().numpy(), lifted.cpu().numpy(), (X, Y, 1))
    assert np.array_equal(want.permute(0, 3, 1, 2).cpu().numpy(), ref)                # a voxel with ~12 600 points among them
    assert torch.equal(plan.lift_splat(prob, ctx), want)
    got_b = plan.pool(lifted.bfloat16())                                               # small integers: exact in bf16
    assert torch.equal(got_b, want)
    got_ob = plan.pool(lifted.bfloat16(), out_bf16_ld=96)
    assert torch.equal(got_ob[..., :C].float(), want.bfloat16().float()) and float(got_ob[..., C:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ level-1 drop-in entry
def _level1(hip, g, f, out, pm, X, Y, Z):
    lib = hip.load()
    B, N, C = f.shape
    hip.check(lib.sgv3d_voxel_pooling_forward(B, N, C, X, Y, Z, g.data_ptr(), f.data_ptr(), out.data_ptr(),
                                              None if pm is None else pm.data_ptr(), hip.stream_handle()), "level1")


def _l1_stats(hip):
    import ctypes
    buf = (ctypes.c_ulonglong * 4)()
    hip.check(hip.load().sgv3d_voxel_pooling_cache_stats(buf), "stats")
    return list(buf)


@pytest.mark.parametrize("misalign,N", [(0, 20011), (1, 20011), (0, 20012)])
def test_level1_entry_accumulates_and_follows_geometry_changes(hip, misalign, N):
    """sgv3d_voxel_pooling_forward as the reference's wrapper uses it: rows are ADDED to output_features and empty voxels
    left alone (atomicAdd semantics, voxel_pooling_forward_cuda.cu:30-33); a changed geom_xyz is served correctly at once
    (gated scatter) and the library-owned plan follows it a call later; pos_memo every time.  ``misalign``: geom_xyz / pos_memo
    at 4-byte aligned addresses only (the prologue's one-point-per-thread form); B * N = 40022 ends in a partial quad of points."""
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")
    rng = np.random.default_rng(21)
    B, C, X, Y, Z = 2, 80, 37, 29, 2
    mk = lambda: rng.integers(-2, 40, size=(B, N, 3)).astype(np.int32)
    ga, gb = mk(), mk()
    ga[..., 2], gb[..., 2] = rng.integers(-1, 3, size=(B, N)), rng.integers(0, 2, size=(B, N))
    feats = rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)
    f = torch.from_numpy(feats).to(DEV)
    refs = {id(g): VPO.forward(g, feats, (X, Y, Z)) for g in (ga, gb)}
    def place(a):
        buf = torch.empty(a.size + 4, dtype=torch.int32, device=DEV)
        view = buf[misalign:misalign + a.size].view(a.shape)
        view.copy_(torch.from_numpy(a))
        assert view.data_ptr() % 16 == 4 * misalign
        return view
    dev = {id(g): place(g) for g in (ga, gb)}
    s0 = _l1_stats(hip)
    for step, g in enumerate((ga, ga, gb, gb, gb, ga, ga, ga)):
        out = torch.ones(B, Y, X, C, device=DEV)                          # NOT zero: the entry accumulates
        pm = place(np.full((B, N, 3), -1, dtype=np.int32))
        _level1(hip, dev[id(g)], f, out, pm, X, Y, Z)
        torch.cuda.synchronize()                                          # (lets the host see the device's note)
        ref_out, ref_pm = refs[id(g)]
        assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), ref_out + 1.0), step
        assert np.array_equal(pm.cpu().numpy().reshape(ref_pm.shape), ref_pm), step
    s1 = _l1_stats(hip)
    assert s1[0] - s0[0] == 8 and s1[1] - s0[1] == 8 and s1[2] == s0[2]   # all eight through the plan entry
    assert s1[3] - s0[3] == 3                                             # first call + one rebuild per geometry change
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")


@pytest.mark.parametrize("C", [80, 20])
def test_level1_fresh_entry_overwrites_garbage_and_follows_geometry_changes(hip, C):
    """sgv3d_voxel_pooling_forward_fresh (what this build's Python operator calls instead of zero fill + level-1 entry): the map
    is WRITTEN whatever it held -- NaN here -- through the first call's build, the steady-state gather, the gated scatter of
    a call whose geom_xyz changed and (C = 20: a channel count the gather does not cover) the scatter-only form."""
    lib = hip.load()
    hip.check(lib.sgv3d_voxel_pooling_cache_clear(), "clear")
    rng = np.random.default_rng(5)
    B, N, X, Y, Z = 2, 15013, 33, 27, 1
    mk = lambda: rng.integers(-2, 36, size=(B, N, 3)).astype(np.int32)
    ga, gb = mk(), mk()
    ga[..., 2], gb[..., 2] = rng.integers(-1, 2, size=(B, N)), 0
    feats = rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)
    f = torch.from_numpy(feats).to(DEV)
    refs = {id(g): VPO.forward(g, feats, (X, Y, Z)) for g in (ga, gb)}
    dev = {id(g): torch.from_numpy(g).to(DEV) for g in (ga, gb)}
    for step, g in enumerate((ga, ga, gb, gb, ga, ga)):
        out = torch.full((B, Y, X, C), float("nan"), device=DEV)
        pm = torch.full((B, N, 3), -1, dtype=torch.int32, device=DEV)
        hip.check(lib.sgv3d_voxel_pooling_forward_fresh(B, N, C, X, Y, Z, dev[id(g)].data_ptr(), f.data_ptr(), out.data_ptr(),
                                                        pm.data_ptr(), hip.stream_handle()), "level1 fresh")
        torch.cuda.synchronize()
        ref_out, ref_pm = refs[id(g)]
        assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), ref_out), step
        assert np.array_equal(pm.cpu().numpy().reshape(ref_pm.shape), ref_pm), step
    hip.check(lib.sgv3d_voxel_pooling_cache_clear(), "clear")


@pytest.mark.parametrize("hot", [False, True])
def test_level1_entry_inside_a_captured_graph(hip, hot):
    """A hipGraph that holds the level-1 call keeps working when geom_xyz is rewritten in place between replays: the capture
    records ONE gated rebuild launch (vp_plan_build_one_kernel: every phase of the plan build behind device-wide barriers), so a
    replay with a changed geom_xyz rebuilds the plan and sums with the deterministic gather in that very replay -- the plan's
    build counter moves, exact sums either way.  ``hot``: voxel populations in every class of the build's sorts, up to one voxel
    with 5 000 points (in-register sorts of 64 .. 2 048, the workgroup sort beyond, in place in global memory past its LDS stage)."""
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")
    rng = np.random.default_rng(33)
    B, N, C, X, Y, Z = (1, 70000, 32, 40, 30, 1) if hot else (1, 9000, 80, 21, 17, 1)

    def geometry(seed):
        r = np.random.default_rng(seed)
        g = r.integers(-1, max(X, Y) + 1, size=(B, N, 3)).astype(np.int32)
        g[..., 2] = 0
        if hot:                                            # populations 5000, 1500, 700, 300, 100 (+ the random background)
            pos = 0
            for k, n in enumerate((5000, 1500, 700, 300, 100)):
                idx = r.permutation(N)[:n]
                g[0, idx, 0], g[0, idx, 1] = 3 + 2 * k + seed % 2, 5 + k
                pos += n
        return g
    g0, g1 = geometry(40), geometry(41)
    feats = rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)
    g = torch.from_numpy(g0).to(DEV)
    f = torch.from_numpy(feats).to(DEV)
    out = torch.zeros(B, Y, X, C, device=DEV)
    s = torch.cuda.Stream()
    s0 = _l1_stats(hip)
    with torch.cuda.stream(s):
        _level1(hip, g, f, out, None, X, Y, Z)                            # eager: creates the plan for this stream
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out.mul_(0.0)                                                 # (a kernel, not a memset node)
            _level1(hip, g, f, out, None, X, Y, Z)
    for cur in (g0, g1, g1, g0, g0):
        g.copy_(torch.from_numpy(cur))
        graph.replay()
        torch.cuda.synchronize()
        ref, _ = VPO.forward(cur, feats, (X, Y, Z))
        assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), ref)
    assert _l1_stats(hip)[3] - s0[3] == 2                                 # the eager build + the one recorded into the graph
    hip.check(hip.load().sgv3d_voxel_pooling_cache_clear(), "clear")


def test_level1_graph_rebuilds_on_two_streams_under_a_saturating_load(hip):
    """Two captured level-1 calls (two streams, two plans) replayed CONCURRENTLY with changed geometry while a third stream keeps every
    compute unit busy with large GEMMs: the one-launch rebuild's grid barrier must not depend on co-residency it was not given.  Its
    grid is sized from the kernel's occupancy, workgroups become resident as the finite load kernels drain, and a barrier that gives
    up after 20 ms leaves through the gated scatter -- so whichever way a replay goes, the sums are exact and the call returns."""
    import time
    lib = hip.load()
    hip.check(lib.sgv3d_voxel_pooling_cache_clear(), "clear")
    B, N, C, X, Y, Z = 1, 120000, 80, 64, 48, 1
    rng = np.random.default_rng(77)
    feats = rng.integers(-4, 5, size=(B, N, C)).astype(np.float32)

    def geometry(seed):
        r = np.random.default_rng(seed)
        g = r.integers(-2, max(X, Y) + 2, size=(B, N, 3)).astype(np.int32)
        g[..., 2] = 0
        return g
    geoms = [geometry(100 + i) for i in range(4)]
    refs = [VPO.forward(g, feats, (X, Y, Z))[0] for g in geoms]
    f = torch.from_numpy(feats).to(DEV)
    lanes = []
    for lane in range(2):
        s = torch.cuda.Stream()
        g = torch.from_numpy(geoms[lane]).to(DEV)
        out = torch.zeros(B, Y, X, C, device=DEV)
        with torch.cuda.stream(s):
            _level1(hip, g, f, out, None, X, Y, Z)                        # eager: this stream's plan
            s.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                out.mul_(0.0)
                _level1(hip, g, f, out, None, X, Y, Z)
        lanes.append((s, g, out, graph))
    load_stream = torch.cuda.Stream()
    a = torch.randn(6144, 6144, device=DEV)
    b = torch.randn(6144, 6144, device=DEV)
    c = torch.empty_like(a)
    torch.cuda.synchronize()
    builds0 = _l1_stats(hip)[3]
    t0 = time.perf_counter()
    for rnd in range(6):
        with torch.cuda.stream(load_stream):
            for _ in range(6):
                torch.matmul(a, b, out=c)                                  # ~3 ms each at the f32 peak: the chip is full throughout
        picks = [(rnd + lane) % 4 for lane in range(2)]
        for (s, g, out, graph), k in zip(lanes, picks):
            with torch.cuda.stream(s):
                g.copy_(torch.from_numpy(geoms[k]).to(DEV), non_blocking=True)      # changed geometry in every replay
                graph.replay()
        torch.cuda.synchronize()
        for (s, g, out, graph), k in zip(lanes, picks):
            assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), refs[k]), (rnd, k)
    wall = time.perf_counter() - t0
    assert wall < 20.0, wall                                               # (a hung barrier used to poll for seconds per phase)
    assert _l1_stats(hip)[3] == builds0                                    # replays record no new builds on the host side
    hip.check(lib.sgv3d_voxel_pooling_cache_clear(), "clear")
