"""Packed weights kept across training steps (sgv3d_amd/pack_cache.py): the forms one gather launch refreshes are bitwise what the
layers' own pack kernels write, the window protocol never serves a stale form, and a step with the cache is the step without it."""
import os

import pytest
import torch

from sgv3d_amd import hip_ops, pack_cache, synthetic
from sgv3d_amd.train_step import DataParallelAdamW

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _setup(seed=0, mixed=True):
    from sgv3d_amd.models.bev_height import BEVHeight
    bconf, hconf = synthetic.small_conf()
    torch.manual_seed(seed)
    model = BEVHeight(bconf, hconf).to(DEV)
    synthetic.randomize_norm_stats_(model, seed=1)
    model.train()
    model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
    imgs = synthetic.make_images(2, final=bconf['final_dim'], device=DEV, seed=0)
    mats = synthetic.make_mats(2, device=DEV, scale=bconf['final_dim'][0] / 864)
    boxes, labels = synthetic.make_gt(2, seed=0, n_range=(10, 40), stress=False)
    boxes, labels = [b.to(DEV) for b in boxes], [l.to(DEV) for l in labels]
    opt = DataParallelAdamW(model.parameters(), lr=2e-3, max_grad_norm=5.0)

    def fb():
        loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats))
        loss.backward()
        return loss
    return model, opt, fb


@pytest.fixture
def mixed_precision():
    saved = hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
    yield
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = saved


def _kernel_counts(fn):
    from torch.profiler import profile, ProfilerActivity
    from torch.autograd import DeviceType
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    counts = {}
    for e in prof.events():
        if e.device_type == DeviceType.CUDA:
            for k in ("pack_weight_kernel", "weight_rot180_transpose_kernel", "patch_pack_kernel", "gather_pack_kernel"):
                if k in e.name:
                    counts[k] = counts.get(k, 0) + 1
    return counts


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_refreshed_forms_are_bitwise_the_pack_kernels_output(mode):
    saved = hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS
    if mode == "bf16":
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
    try:
        model, opt, fb = _setup()
        assert pack_cache.ENABLED
        for _ in range(3):
            opt.zero_grad()
            fb()
            opt.step()
        opt.zero_grad()                                   # refreshes every registered form from the weights of step 3
        cache = opt.packs
        jobs = [(e, n, j) for e in cache.entries.values() if e.tracked for n, j in e.jobs.items()]
        assert len(jobs) >= (20 if mode == "bf16" else 8), len(jobs)      # (f32: the 3x3 layers read Winograd forms and pack per call)
        kinds = {n for _, n, _ in jobs}
        assert 'w' in kinds, kinds            # (which bf16 forms appear depends on the per-layer choices: test_every_permutation_form...)
        for e, name, j in jobs:
            want = j.getter(e.make(e.param.detach()))     # the layer's own pack path on the current weights
            want = want.view(torch.bfloat16) if want.dtype == torch.uint8 else want
            got = j.dst
            assert torch.equal(got.view(torch.int16 if j.bf16 else torch.int32), want.reshape(-1).view(torch.int16 if j.bf16 else torch.int32)), name
            # and the map is a permutation with padding: every parameter element appears, or the form is a sub-kernel (stride-2 phases)
            assert int(j.idx.max()) < e.param.numel() and int(j.idx.min()) >= -1
        # every form a kept object HOLDS -- registered or not (a 1x1 layer's packed rows are a view of its source) -- is current
        held = 0
        for e in cache.entries.values():
            if not e.tracked or e.pc is None:
                continue
            fresh = e.make(e.param.detach())
            for attr, get in (("_w", lambda pc: pc.w), ("w_patch", lambda pc: pc._patch_weights()), ("w_dw", lambda pc: pc._dw_weights()),
                              ("w_bf16", lambda pc: pc._bf16_weights())):
                have = getattr(e.pc, attr, None)
                if have is not None:
                    want = get(fresh)
                    assert torch.equal(have.reshape(-1).view(torch.uint8), want.reshape(-1).view(torch.uint8)), (attr, tuple(e.param.shape))
                    held += 1
        assert held >= len(jobs)
        torch.cuda.synchronize()
    finally:
        hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = saved


def test_a_step_with_the_cache_launches_no_per_layer_pack_and_equals_the_step_without_it(mixed_precision):
    """After two optimiser steps (the kept forms have been refreshed twice): forward + backward of the third step through the kept
    forms against the same step with every layer packing per call, FROM THE SAME STATE -- the loss is bitwise the same (the kernels
    read the same bytes), the gradients differ by the float atomics of the deformable-convolution adjoint only.  And the step with
    the cache launches one gather instead of the per-layer pack / rotate kernels."""
    model, opt, fb = _setup()
    for _ in range(2):
        opt.zero_grad()
        fb()
        opt.step()
    stats = {k: v.clone() for k, v in model.state_dict().items() if 'running_' in k or 'num_batches' in k}

    def run(enabled):
        model.load_state_dict(stats, strict=False)           # (BatchNorm statistics move in a training forward)
        opt.zero_grad()
        pack_cache.ENABLED = enabled
        try:
            torch.manual_seed(11)                            # dropout
            loss = fb().detach().clone()
        finally:
            pack_cache.ENABLED = True
        return loss, torch.cat([g.clone() for _, g, _ in opt.flat.buckets])
    la, ga = run(True)
    lb, gb = run(False)
    lc, gc = run(False)
    assert torch.equal(la, lb), (la, lb)
    noise = float((gb - gc).abs().max())                      # two runs of the SAME path: the atomics' rounding noise
    assert float((ga - gb).abs().max()) <= max(4 * noise, 1e-6 * float(gb.abs().max())), (float((ga - gb).abs().max()), noise)

    def one():
        opt.zero_grad()
        fb()
        opt.step()
    c1 = _kernel_counts(one)
    pack_cache.ENABLED = False
    try:
        c0 = _kernel_counts(one)
    finally:
        pack_cache.ENABLED = True
    print("launches per step with / without the cache:", c1, c0)
    per_layer = lambda c: sum(v for k, v in c.items() if k != "gather_pack_kernel")
    assert c1.get("gather_pack_kernel", 0) == 1 and c0.get("gather_pack_kernel", 0) == 0
    assert per_layer(c0) >= 40 and per_layer(c1) <= per_layer(c0) // 8, (c1, c0)


def test_outside_the_window_every_layer_packs_from_the_current_weights(mixed_precision):
    """Weights written behind torch's back (through the flat buckets) after a step, and a forward without ``opt.zero_grad()``: the
    closed window makes every layer pack per call, so the forward sees the new weights; the next ``zero_grad`` refreshes the kept
    forms from them too."""
    model, opt, fb = _setup()
    for _ in range(2):
        opt.zero_grad()
        fb()
        opt.step()
    assert not opt.packs.open
    with torch.no_grad():
        for p, _, _ in opt.flat.buckets:
            p.mul_(1.25)                                   # no parameter's _version moves
    imgs = synthetic.make_images(2, final=synthetic.small_conf()[0]['final_dim'], device=DEV, seed=0)
    mats = synthetic.make_mats(2, device=DEV, scale=synthetic.small_conf()[0]['final_dim'][0] / 864)

    def forward():
        with torch.no_grad():
            return [{k: v.clone() for k, v in t[0].items()} for t in model(imgs, mats)]
    torch.manual_seed(3)
    closed = forward()
    pack_cache.ENABLED = False
    try:
        torch.manual_seed(3)
        plain = forward()
    finally:
        pack_cache.ENABLED = True
    opt.zero_grad()                                        # window open: kept forms, refreshed from the scaled weights
    torch.manual_seed(3)
    opened = forward()
    for a, b, c in zip(closed, plain, opened):
        for k in a:
            assert torch.equal(a[k], b[k]), k
            assert torch.equal(c[k], b[k]), k


@pytest.mark.parametrize("shape,kw", [
    ((64, 64, 3, 3), dict(pad=1)),                    # 3x3: implicit-GEMM rows, bf16 patch fragments, direct-weight fragments
    ((96, 160, 1, 1), dict()),                        # 1x1 with padded output rows
    ((128, 32, 2, 2), dict(stride=2, transposed=True)),
])
def test_every_permutation_form_is_refreshed_bitwise(shape, kw):
    """Every packed form the cache keeps (implicit-GEMM rows, their bf16 copy, the bf16 fragment orders of the patch and the
    direct-weight kernels), made through a cache entry, then the parameter changed behind torch's back and ONE gather launch: bitwise
    what the pack kernels write for the new weights -- also through a rotated / transposed source (the data-gradient form)."""
    from sgv3d_amd.hip_ops import PackedConv
    from sgv3d_amd.conv_grad import _rot180_transpose
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.nn.Parameter(torch.randn(shape, generator=g).to(DEV))
    cache = pack_cache.PackCache([w])
    cache.open = True
    transposed = kw.get('transposed', False)
    cin = shape[0] if transposed else shape[1]
    makes = {"plain": lambda v: PackedConv(v, cin_pad=cin, **kw)}
    if not transposed:
        makes["rotated"] = lambda v: PackedConv(_rot180_transpose(v), cin_pad=shape[0], pad_out=True, **kw)
    getters = {"w": lambda pc: pc.w, "w_bf16": lambda pc: pc._bf16_weights(), "w_dw": lambda pc: pc._dw_weights()}
    if shape[2] == 3 and not transposed:
        getters["w_patch"] = lambda pc: pc._patch_weights()
    entries = {}
    for name, make in makes.items():
        e = entries[name] = cache.lookup(w, name, make)
        pc = e.conv()
        for get in getters.values():
            if name == "rotated" and get is getters.get("w_patch") and not pc.patch_ok:
                continue
            get(pc)
    assert sum(len(e.jobs) for e in entries.values()) >= len(getters)
    with torch.no_grad():
        w.data.mul_(-0.37).add_(0.01)                    # (through .data: no version bump, like the fused AdamW)
    cache.refresh(DEV)
    for name, e in entries.items():
        fresh = e.make(w.detach())
        for fname, job in e.jobs.items():
            want = getters[fname](fresh)
            assert torch.equal(job.dst.view(torch.uint8), want.reshape(-1).view(torch.uint8)), (name, fname)
        assert e.pc is not None and all(getattr(e.pc, a, None) is not None for a in ("_w",))
