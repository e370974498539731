"""Optimiser step on the MI355X: the fused AdamW kernel over flat buckets against
torch.optim.AdamW, and a small end-to-end training slice through the HIP convolution autograd functions."""
import pytest
import torch

from sgv3d_amd import conv_grad
from sgv3d_amd.train_step import DataParallelAdamW

pytestmark = pytest.mark.gpu


def _nets():
    torch.manual_seed(0)
    make = lambda: torch.nn.Sequential(torch.nn.Conv2d(4, 16, 3, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                                       torch.nn.Conv2d(16, 7, 1)).cuda()
    a = make()
    b = make()
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize("bucket_bytes", [512, 256 << 20])
def test_fused_adamw_tracks_torch_adamw(bucket_bytes):
    ours, ref = _nets()
    opt = DataParallelAdamW(ours.parameters(), lr=3e-3, weight_decay=1e-2, bucket_bytes=bucket_bytes)
    ropt = torch.optim.AdamW(ref.parameters(), lr=3e-3, weight_decay=1e-2)
    g = torch.Generator(device='cuda').manual_seed(1)
    for it in range(12):
        x = torch.randn(3, 4, 12, 12, device='cuda', generator=g)
        lr = 3e-3 * (0.1 if it >= 8 else 1.0)
        for group in ropt.param_groups:
            group['lr'] = lr
        opt.zero_grad()
        ropt.zero_grad()
        ours(x).square().mean().backward()
        ref(x).square().mean().backward()
        opt.step(lr)
        ropt.step()
    for p, q in zip(ours.parameters(), ref.parameters()):
        err = float((p.detach() - q.detach()).abs().max())
        assert err <= 1e-4 * max(1.0, float(q.detach().abs().max())), err      # 12 steps of lr 3e-3: a wrong rule is off by 1e-2


def test_fused_adamw_same_gradients_match_to_rounding():
    """Identical gradient streams into the kernel and into torch.optim.AdamW: parameters agree to fp32 rounding."""
    g = torch.Generator(device='cuda').manual_seed(5)
    p = torch.randn(100003, device='cuda', generator=g).requires_grad_(True)
    q = p.detach().clone().requires_grad_(True)
    opt = DataParallelAdamW([p], lr=2e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    ropt = torch.optim.AdamW([q], lr=2e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    for it in range(20):
        grad = torch.randn(100003, device='cuda', generator=g) * (10.0 ** (it % 5 - 2))
        p.grad.copy_(grad)
        q.grad = grad.clone()
        opt.step()
        ropt.step()
    err = float((p.detach() - q.detach()).abs().max())
    assert err <= 6e-6, err          # |p| up to 4: a few ulps after 20 steps


def test_adamw_with_device_scalars_is_bitwise_the_host_scalar_update():
    """``step(recorded=True)`` after ``stage_hyper()`` (sgv3d_adamw_step_dev: lr and the two bias corrections read from device memory,
    what a recorded hipGraph launch needs) against ``step()``: the same bits after every step, with a changing learning rate."""
    g = torch.Generator(device='cuda').manual_seed(6)
    p = torch.randn(70001, device='cuda', generator=g).requires_grad_(True)
    q = p.detach().clone().requires_grad_(True)
    a = DataParallelAdamW([p], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7)
    b = DataParallelAdamW([q], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7)
    for it in range(12):
        grad = torch.randn(70001, device='cuda', generator=g) * (10.0 ** (it % 4 - 2))
        lr = 2e-3 * (0.5 if it >= 6 else 1.0)
        p.grad.copy_(grad)
        q.grad.copy_(grad)
        a.step(lr)
        b.stage_hyper(lr)
        b.step(lr, recorded=True)
        assert a.steps == b.steps == it + 1
        assert torch.equal(p.detach(), q.detach()), it
        assert all(torch.equal(x, y) for sa, sb in zip(a.state, b.state) for x, y in zip(sa, sb))


def test_clipped_update_is_clip_grad_norm_then_adamw():
    """``max_grad_norm`` (Lightning's gradient_clip_val=5, exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:405) against
    ``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.AdamW`` on the same gradients: several buckets (the norm is global), steps whose
    norm is above the bound (clipped) and below it (coefficient exactly 1), host-scalar and device-scalar update entries."""
    g = torch.Generator(device='cuda').manual_seed(7)
    shapes = [(257, 33), (4096,), (64, 3, 7, 7), (1000003,), (5,)]
    for recorded in (False, True):
        ps = [torch.randn(*sh, device='cuda', generator=g).requires_grad_(True) for sh in shapes]
        qs = [p.detach().clone().requires_grad_(True) for p in ps]
        opt = DataParallelAdamW(ps, lr=2e-3, betas=(0.9, 0.99), weight_decay=0.01, bucket_bytes=1 << 20, max_grad_norm=5.0)
        assert len(opt.flat.buckets) >= 3
        ropt = torch.optim.AdamW(qs, lr=2e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
        for it in range(10):
            scale = [1.0, 1e-4, 30.0, 3e-3][it % 4]                 # norms ~1e3, ~0.1, ~3e4, ~3: clipped / not / clipped / not
            for p, q in zip(ps, qs):
                grad = torch.randn(p.shape, device='cuda', generator=g) * scale
                p.grad.copy_(grad)
                q.grad = grad.clone()
            want_norm = float(torch.nn.utils.clip_grad_norm_(qs, 5.0))
            ropt.step()
            if recorded:
                opt.stage_hyper()
            opt.step(recorded=recorded)
            assert abs(opt.grad_norm() - want_norm) <= 2e-6 * want_norm, (it, opt.grad_norm(), want_norm)
            coef = opt.clip_coefficient()
            assert (coef == 1.0) if want_norm < 5.0 else abs(coef - 5.0 / (want_norm + 1e-6)) <= 1e-6 * coef, (it, coef, want_norm)
        for p, q in zip(ps, qs):
            err = float((p.detach() - q.detach()).abs().max())
            assert err <= 1e-5, (recorded, tuple(p.shape), err)       # |p| up to ~5, ten steps of lr 2e-3


def test_clipping_changes_the_update_and_off_is_bitwise_the_old_path():
    """max_grad_norm=None launches nothing new (bitwise the unclipped update); with a bound below the norm the step shrinks."""
    g = torch.Generator(device='cuda').manual_seed(8)
    base = torch.randn(50000, device='cuda', generator=g)
    grad = torch.randn(50000, device='cuda', generator=g) * 10
    outs = []
    for mg in (None, 0, 1e9, 1.0):
        p = base.clone().requires_grad_(True)
        opt = DataParallelAdamW([p], lr=1e-2, weight_decay=0.0, eps=1e-3, max_grad_norm=mg)
        p.grad.copy_(grad)
        opt.step()
        outs.append(p.detach().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])         # coefficient exactly 1
    assert not torch.equal(outs[0], outs[3])


def test_frozen_stem_gets_no_gradient_and_keeps_its_statistics():
    """``frozen_stages=0`` (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:48; mmdet 2.19.0 ResNet._freeze_stages):
    after a training step of the small model the image backbone's conv1 / bn1 have no gradient, are in no optimiser bucket, did not
    move, bn1's running statistics and batch counter are untouched -- while the first trainable layer got a gradient and its
    BatchNorm's statistics moved.  ``train()`` re-freezes after ``eval()``; ``frozen_stages=-1`` trains the stem."""
    from sgv3d_amd import synthetic
    from sgv3d_amd.models.bev_height import BEVHeight
    dev = torch.device("cuda", 0)
    bconf, hconf = synthetic.small_conf()
    assert bconf['img_backbone_conf']['frozen_stages'] == 0
    for frozen in (0, -1):
        torch.manual_seed(0)
        bc = dict(bconf, img_backbone_conf=dict(bconf['img_backbone_conf'], frozen_stages=frozen))
        model = BEVHeight(bc, hconf).to(dev)
        synthetic.randomize_norm_stats_(model, seed=1)
        model.eval().train()                                             # (train() has to re-apply the freeze)
        r = model.backbone.img_backbone
        model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
        assert r.bn1.training == (frozen < 0) and r.conv1.weight.requires_grad == (frozen < 0) and r.bn1.weight.requires_grad == (frozen < 0)
        assert r.layer1[0].bn1.training and r.layer1[0].conv1.weight.requires_grad
        imgs = synthetic.make_images(2, final=bc['final_dim'], device=dev, seed=0)
        mats = synthetic.make_mats(2, device=dev, scale=bc['final_dim'][0] / 864)
        boxes, labels = synthetic.make_gt(2, seed=0, n_range=(10, 40), stress=False)
        opt = DataParallelAdamW(model.parameters(), lr=2e-4, max_grad_norm=5.0)
        in_buckets = {id(p) for _, _, entries in opt.flat.buckets for p, _, _ in entries}
        before = {k: v.detach().clone() for k, v in r.state_dict().items() if k.startswith(('conv1.', 'bn1.', 'layer1.0.bn1.', 'layer1.0.conv1.'))}
        opt.zero_grad()
        loss = model.loss(model.get_targets([b.to(dev) for b in boxes], [l.to(dev) for l in labels]), model(imgs, mats))
        loss.backward()
        g1 = r.layer1[0].conv1.weight.grad
        assert g1 is not None and float(g1.abs().max()) > 0
        if frozen >= 0:
            assert r.conv1.weight.grad is None and r.bn1.weight.grad is None and r.bn1.bias.grad is None
            assert id(r.conv1.weight) not in in_buckets and id(r.bn1.weight) not in in_buckets
        else:
            assert r.conv1.weight.grad is not None and float(r.conv1.weight.grad.abs().max()) > 0
        opt.step()
        torch.cuda.synchronize()
        after = r.state_dict()
        for k, v in before.items():
            same = torch.equal(v, after[k])
            if k.startswith(('conv1.', 'bn1.')):
                assert same == (frozen >= 0), (frozen, k)
            else:
                assert not same, (frozen, k)                            # (weights, statistics and the counter of layer1.0 all move)
        assert opt.grad_norm() > 0


def test_graph_replays_back_to_back_follow_the_eager_steps():
    """Several replays of the recorded step with NO synchronisation in between (tools/train_bench.py --graph times them like that)
    against the same number of eager steps from the same state, early in training where the bias corrections change quickly: the
    per-step scalars travel as kernel arguments of ``sgv3d_adamw_set_hyper``, so staging step t + k cannot leak into step t (a reused
    pinned staging buffer could).  Clipping (max_grad_norm=5) is part of the recorded step."""
    from sgv3d_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    w1 = (torch.randn(32, 8, 3, 3) * 0.2).cuda().requires_grad_(True)
    w2 = (torch.randn(4, 32, 1, 1) * 0.2).cuda().requires_grad_(True)
    x = torch.randn(2, 16, 20, 8, device=dev)
    target = torch.randn(2, 16, 20, 4, device=dev) * 30
    opt = DataParallelAdamW([w1, w2], lr=1e-2, weight_decay=0.0, max_grad_norm=5.0)

    def forward_backward():
        y = conv_grad.conv2d(torch.relu(conv_grad.conv2d(x, w1, None, 1, 1, 1)), w2)
        loss = (y - target).square().mean()
        loss.backward()
        return loss

    def eager():
        opt.zero_grad()
        forward_backward()
        opt.step()

    eager()                                                               # (per-layer kernel choices)
    snap = [p.clone() for p, _, _ in opt.flat.buckets], [(m.clone(), v.clone()) for m, v in opt.state]

    def restore():
        for (p, _, _), q in zip(opt.flat.buckets, snap[0]):
            p.copy_(q)
        for (m, v), (m0, v0) in zip(opt.state, snap[1]):
            m.copy_(m0); v.copy_(v0)
        opt.steps = 0                                                     # the first steps: 1 - 0.9^t moves by a factor 1.9 from t=1 to 2

    restore()
    for _ in range(8):
        eager()
    torch.cuda.synchronize()
    want = torch.cat([p for p, _, _ in opt.flat.buckets]).clone()
    restore()
    graphed = GraphedTrainStep(forward_backward, opt, warmup=0, strict=True)
    assert graphed.graph is not None and graphed.in_graph_update
    restore()
    for _ in range(8):
        graphed()                                                         # no sync, no .item()
    torch.cuda.synchronize()
    got = torch.cat([p for p, _, _ in opt.flat.buckets])
    assert opt.steps == 8
    assert opt.clip_coefficient() < 1.0                                   # the clip was active
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-7, float((got - want).abs().max())


_DP_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=2)
try:
    probe = torch.ones(4, device=dev)
    dist.all_reduce(probe)
    assert float(probe[0]) == 2.0
except Exception as e:                                   # a gloo build without device-tensor support
    if rank == 0:
        print(json.dumps({"skip": f"gloo cannot all-reduce device tensors here: {type(e).__name__}: {e}"}))
    sys.exit(0)
from sgv3d_amd import synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.train_step import DataParallelAdamW
bconf, hconf = synthetic.small_conf(depth=18)
torch.manual_seed(3 + rank)                                # different initial weights per rank: the constructor broadcasts rank 0's
model = BEVHeight(bconf, hconf).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
stem = model.backbone.img_backbone
# the frozen stem is not broadcast (in the reference it comes from the checkpoint every rank loads): make it equal by hand
for t in list(stem.conv1.parameters()) + list(stem.bn1.parameters()) + list(stem.bn1.buffers()):
    dist.broadcast(t.data, src=0)
stem_before = [t.detach().clone() for t in list(stem.conv1.parameters()) + list(stem.bn1.parameters()) + list(stem.bn1.buffers())]
opt = DataParallelAdamW(model.parameters(), lr=2e-3, max_grad_norm=5.0, bucket_bytes=4 << 20)
assert len(opt.flat.buckets) >= 4 and opt._collectives()
opt.check_replicas()
imgs = synthetic.make_images(2, final=bconf['final_dim'], device=dev, seed=10 + rank)     # different data per rank
mats = synthetic.make_mats(2, device=dev, scale=bconf['final_dim'][0] / 864)
boxes, labels = synthetic.make_gt(2, seed=20 + rank, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
opt.zero_grad()
loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats))
loss.backward()
assert stem.conv1.weight.grad is None and stem.bn1.weight.grad is None
local = torch.cat([g for _, g, _ in opt.flat.buckets]).clone()
before = torch.cat([p for p, _, _ in opt.flat.buckets]).clone()
both = [torch.empty_like(local) for _ in range(2)]
dist.all_gather(both, local)
avg = (both[0] + both[1]) / 2
want_norm = float(avg.double().norm())
opt.all_reduce_grads()
opt.step()
torch.cuda.synchronize()
summed = torch.cat([g for _, g, _ in opt.flat.buckets])
assert torch.allclose(summed, both[0] + both[1], rtol=1e-6, atol=1e-7)
assert abs(opt.grad_norm() - want_norm) <= 1e-5 * want_norm, (opt.grad_norm(), want_norm)
coef = min(1.0, 5.0 / (want_norm + 1e-6))
assert coef < 1.0, want_norm                              # an untrained detector: the clip is active
# the first AdamW step from zero moments moves every parameter by lr * g / (|g| + eps) -- sign-like --, so check the update with
# torch's own AdamW on the clipped average
q = before.clone().requires_grad_(True)
q.grad = avg * coef
ropt = torch.optim.AdamW([q], lr=2e-3, weight_decay=1e-7)
ropt.step()
after = torch.cat([p for p, _, _ in opt.flat.buckets])
err = float((after - q.detach()).abs().max())
opt.check_replicas()                                       # both ranks applied the same update
assert all(torch.equal(a, b) for a, b in zip(stem_before, list(stem.conv1.parameters()) + list(stem.bn1.parameters()) + list(stem.bn1.buffers())))
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"ok": True, "grad_norm": want_norm, "coef": coef, "update_err": err, "buckets": len(opt.flat.buckets)}))
"""


def test_two_rank_clipped_frozen_stem_step(tmp_path):
    """The data-parallel step of BASELINE configs[3] with the reference's two training semantics -- ``frozen_stages=0`` and
    ``gradient_clip_val=5`` -- on TWO ranks sharing this box's one MI355X (gloo carries the device tensors; RCCL refuses two ranks on
    one device): the buckets hold the sum of both ranks' gradients, the norm is the averaged gradient's, the clipped update is
    torch.optim.AdamW's on ``avg * min(1, 5 / (norm + 1e-6))``, both ranks end with identical parameters, and the frozen stem has no
    gradient, sits in no bucket and does not move."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % root)
    env = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST",)}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-3000:]
    rec = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    if "skip" in rec:
        pytest.skip(rec["skip"])
    assert rec["ok"] and rec["update_err"] <= 2e-6, rec
    print(rec)


def test_graphed_train_step_is_the_eager_step():
    """train_step.GraphedTrainStep -- zero_grad + forward + targets + loss + backward + AdamW of the small model recorded as one hipGraph --
    against the eager step FROM THE SAME STATE (parameters, AdamW moments, step counter restored in between): the same loss bit for
    bit, every gradient and every updated parameter within the run-to-run noise of the eager step itself (the deformable-convolution
    adjoint adds with float atomics), on the first replay and on a later one (the step counter and the bias corrections follow)."""
    from sgv3d_amd import synthetic
    from sgv3d_amd.models.bev_height import BEVHeight
    from sgv3d_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda", 0)
    bconf, hconf = synthetic.small_conf()
    torch.manual_seed(0)
    model = BEVHeight(bconf, hconf).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
    imgs = synthetic.make_images(2, final=bconf['final_dim'], device=dev, seed=0)
    mats = synthetic.make_mats(2, device=dev, scale=bconf['final_dim'][0] / 864)
    boxes, labels = synthetic.make_gt(2, seed=0, n_range=(10, 40), stress=False)
    boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
    opt = DataParallelAdamW(model.parameters(), lr=2e-4, max_grad_norm=5.0)      # (the reference's gradient_clip_val: part of the recorded step)

    def forward_backward():
        loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats))
        loss.backward()
        return loss

    def eager():
        opt.zero_grad()
        loss = forward_backward()
        opt.step()
        return float(loss.detach())

    def snapshot():
        return [p.clone() for p, _, _ in opt.flat.buckets], [(m.clone(), v.clone()) for m, v in opt.state], opt.steps

    def restore(snap):
        for (p, _, _), q in zip(opt.flat.buckets, snap[0]):
            p.copy_(q)
        for (m, v), (m0, v0) in zip(opt.state, snap[1]):
            m.copy_(m0); v.copy_(v0)
        opt.steps = snap[2]

    def state():
        torch.cuda.synchronize()
        return torch.cat([g for _, g, _ in opt.flat.buckets]).clone(), torch.cat([p for p, _, _ in opt.flat.buckets]).clone()

    for _ in range(3):
        eager()
    snap = snapshot()
    runs = {}
    for tag in ("eager_a", "eager_b"):
        restore(snap)
        runs[tag] = (eager(),) + state()
    restore(snap)
    graphed = GraphedTrainStep(forward_backward, opt, warmup=0, strict=True)
    assert graphed.graph is not None and graphed.in_graph_update and opt.steps == snap[2]
    for tag in ("graph_a", "graph_b"):
        restore(snap)
        runs[tag] = (float(graphed().detach()),) + state()
        assert opt.steps == snap[2] + 1
    assert runs["eager_a"][0] == runs["eager_b"][0] == runs["graph_a"][0] == runs["graph_b"][0]        # the forward is deterministic
    gscale, pscale = float(runs["eager_a"][1].abs().max()), float(runs["eager_a"][2].abs().max())
    noise_g = float((runs["eager_a"][1] - runs["eager_b"][1]).abs().max())
    noise_p = float((runs["eager_a"][2] - runs["eager_b"][2]).abs().max())
    for tag in ("graph_a", "graph_b"):
        dg = float((runs[tag][1] - runs["eager_a"][1]).abs().max())
        dp = float((runs[tag][2] - runs["eager_a"][2]).abs().max())
        assert dg <= max(4 * noise_g, 1e-5 * gscale), (tag, dg, noise_g, gscale)
        assert dp <= max(4 * noise_p, 1e-5 * pscale), (tag, dp, noise_p, pscale)
    # a later replay: two more steps each way from the same state (step counter, bias corrections, moments carried by the replays)
    restore(snap)
    le = [eager() for _ in range(3)]
    pe = state()[1]
    restore(snap)
    lg = [float(graphed().detach()) for _ in range(3)]
    pg = state()[1]
    assert opt.steps == snap[2] + 3
    assert max(abs(a - b) / abs(a) for a, b in zip(le, lg)) <= 1e-3, (le, lg)
    assert float((pe - pg).abs().max()) <= max(50 * noise_p, 1e-4 * pscale)
    print(f"graphed step against the eager step: loss {runs['graph_a'][0]:.6f} (identical); gradients {noise_g:.1e} eager-vs-eager, "
          f"{float((runs['graph_a'][1] - runs['eager_a'][1]).abs().max()):.1e} graph-vs-eager of {gscale:.2e}; three steps {le} / {lg}")


def test_graphed_train_step_through_train_bench():
    """The same through tools/train_bench.py --graph (what the profiles are made with): the constructor's eager step and the replays
    add up to the eager run's optimiser steps; the last step's loss agrees with the eager run's to the sensitivity of the
    trajectory (a training trajectory amplifies the atomics' rounding noise; eager runs differ from each other by as much)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "tools", "train_bench.py"), "--config", "small", "--batch", "2", "--steps", "4", "--no-dropout"]
    eager = _run_json(cmd + ["--warmup", "4"], base)                       # 4 + 4 steps
    graph = _run_json(cmd + ["--warmup", "2", "--graph"], base)            # 2 eager + 1 in the constructor + 1 + 4 replays
    assert graph["graph"] is True and graph["graph_replays"] == 5 and graph["update_in_graph"] is True
    assert eager["optimizer_steps"] == graph["optimizer_steps"] == 8
    assert abs(graph["loss"] - eager["loss"]) <= 1e-2 * abs(eager["loss"]), (graph["loss"], eager["loss"])
    assert abs(graph["param_checksum"] - eager["param_checksum"]) <= 1e-5 * eager["param_checksum"]
    print(f"small model, 8 steps: loss {eager['loss']:.5f} eager / {graph['loss']:.5f} graphed")


def test_training_slice_reduces_the_loss():
    """conv -> relu -> conv regression trained for a few steps entirely on the HIP kernels (forward, data and weight
    gradients, fused AdamW); an identical torch model with torch.optim.AdamW must follow the same trajectory."""
    torch.manual_seed(3)
    w1 = (torch.randn(32, 8, 3, 3) * 0.2).cuda().requires_grad_(True)
    b1 = torch.zeros(32).cuda().requires_grad_(True)
    w2 = (torch.randn(4, 32, 1, 1) * 0.2).cuda().requires_grad_(True)
    ref = [t.detach().clone().requires_grad_(True) for t in (w1, b1, w2)]
    opt = DataParallelAdamW([w1, b1, w2], lr=1e-2, weight_decay=0.0)
    ropt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.0)
    x = torch.randn(2, 16, 20, 8, device='cuda')                 # NHWC
    target = torch.randn(2, 16, 20, 4, device='cuda')
    losses, rlosses = [], []
    for _ in range(15):
        opt.zero_grad()
        y = conv_grad.conv2d(torch.relu(conv_grad.conv2d(x, w1, b1, 1, 1, 1)), w2)
        loss = (y - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        ropt.zero_grad()
        xr = x.permute(0, 3, 1, 2)
        yr = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(xr, ref[0], ref[1], 1, 1)), ref[2])
        rl = (yr.permute(0, 2, 3, 1) - target).square().mean()
        rl.backward()
        ropt.step()
        rlosses.append(float(rl.detach()))
    assert losses[-1] < 0.8 * losses[0]
    assert all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(losses, rlosses)), (losses, rlosses)


def test_gradient_slots_write_into_the_buckets():
    """Weight and BatchNorm gradients land in the flat buckets without an accumulate launch (grad_slots): from the second
    step on ``p.grad`` is the kernel's own output view of the bucket, with the bits of the accumulate path (0 + g == g), and a
    second backward before the step still accumulates."""
    from sgv3d_amd import grad_slots, train_step
    from sgv3d_amd.norm_grad import batch_norm_act

    def run(direct):
        old = train_step.DIRECT_GRADS
        train_step.DIRECT_GRADS = direct
        try:
            torch.manual_seed(11)
            w1 = (torch.randn(32, 8, 3, 3) * 0.2).cuda().requires_grad_(True)
            w2 = (torch.randn(8, 32, 1, 1) * 0.2).cuda().requires_grad_(True)
            wt = (torch.randn(8, 4, 2, 2) * 0.2).cuda().requires_grad_(True)          # transposed convolution [cin, cout, k, k]
            bn = torch.nn.BatchNorm2d(32).cuda().train()
            params = [w1, w2, wt, bn.weight, bn.bias]
            opt = DataParallelAdamW(params, lr=1e-2, weight_decay=0.0)
            x = torch.randn(2, 10, 12, 8, device='cuda')
            grads, aliased = [], []
            for it in range(3):
                opt.zero_grad()
                for rep in range(2 if it == 2 else 1):                                 # the last step: two backwards, accumulated
                    h = batch_norm_act(bn, conv_grad.conv2d(x, w1, None, 1, 1, 1), relu=True)
                    y = conv_grad.conv_transpose2d(conv_grad.conv2d(h, w2), wt, 2)
                    y.square().mean().backward()
                opt.flat.check_views()
                aliased.append([p.grad is not None and int(p.data_ptr()) in grad_slots.CAPABLE for p in params])
                grads.append([p.grad.detach().clone() for p in params])
                opt.step()
            return grads, aliased
        finally:
            train_step.DIRECT_GRADS = old

    g1, a1 = run(True)
    g0, _ = run(False)
    assert all(a1[1]) and all(a1[2])                                                   # every parameter here has a slot-aware producer
    for step1, step0 in zip(g1, g0):
        for a, b in zip(step1, step0):
            assert torch.equal(a, b)


def test_adamw_bandwidth_smoke():
    """64 Mi parameters through the fused update; prints the achieved HBM rate (28 bytes per parameter)."""
    n = 64 << 20
    p = torch.randn(n, device='cuda').requires_grad_(True)
    opt = DataParallelAdamW([p], lr=1e-3)
    p.grad.normal_()
    opt.step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        opt.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(5))[2]
    print(f"fused AdamW: {n * 28 / us / 1e6:.2f} TB/s ({us:.0f} us for {n >> 20} Mi parameters)")
    assert n * 28 / us / 1e6 > 2.0


# ------------------------------------------------------------------------------------------------ BASELINE configs[3] stand-in
def _run_json(cmd, env, timeout=900):
    import json
    import subprocess
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_cfg4_share_training_step_through_one_rank_rccl():
    """BASELINE configs[3] (global batch 32 over 8 MI355X) cannot run on one GPU; its per-GPU share can: the cfg-2 model at
    batch 4, one FULL data-parallel step -- parameter broadcast, bucket all-reduces launched from inside backward, fused
    AdamW -- through a real RCCL process group of ONE rank (SGV3D_FORCE_DIST=1).  A 1-rank SUM all-reduce is the identity,
    so the run must land where the same steps without any process group land."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "train_bench.py"), "--config", "cfg4", "--steps", "2", "--warmup", "1"]
    base = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env = dict(base, SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist = _run_json(cmd, env)
    assert dist["backend"] == "nccl" and dist["world_size"] == 1 and dist["collectives_active"] is True
    assert dist["batch_per_gpu"] == 4 and dist["allreduce_buckets"] >= 1 and dist["allreduce_bytes_per_step"] > 250e6
    assert dist["allreduces_launched_inside_backward"] == dist["allreduce_buckets"]      # overlapped with backward
    assert dist["loss"] == dist["loss"] and abs(dist["loss"]) < 1e6
    plain = _run_json(cmd, base)
    assert plain["collectives_active"] is False and plain["world_size"] == 1
    # deformable-conv input gradients use float atomics (order-nondeterministic), so not bitwise: 3 AdamW steps of lr 1.25e-5
    assert abs(dist["loss"] - plain["loss"]) <= 2e-3 * max(1.0, abs(plain["loss"])), (dist["loss"], plain["loss"])
    assert abs(dist["param_checksum"] - plain["param_checksum"]) <= 1e-5 * plain["param_checksum"]
    print(f"cfg-4 share through 1-rank RCCL: {dist['ms_per_step']:.1f} ms / step, {dist['allreduce_bytes_per_step'] / 1e6:.0f} MB "
          f"all-reduced in {dist['allreduce_buckets']} buckets; without a process group {plain['ms_per_step']:.1f} ms")


def test_cfg4_share_mixed_precision_step_through_one_rank_rccl():
    """The same per-GPU share of BASELINE configs[3] as a MIXED-PRECISION step (``--dtype bf16``: every convolution product --
    forward, data gradient, weight gradient -- on the bf16 matrix cores, f32 master weights / gradients / AdamW; the reference's
    documented training command uses --amp_backend native, docs/run_and_eval.md:5,16): through the 1-rank RCCL group, finite,
    and within 2 % of the f32 step's loss after the same three steps on the same data."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env = dict(base, SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29536")
    cmd = [sys.executable, os.path.join(root, "tools", "train_bench.py"), "--config", "cfg4", "--steps", "2", "--warmup", "1"]
    mixed = _run_json(cmd + ["--dtype", "bf16"], env)
    assert mixed["dtype"] == "bf16" and mixed["backend"] == "nccl" and mixed["collectives_active"] is True and mixed["batch_per_gpu"] == 4
    assert mixed["allreduces_launched_inside_backward"] == mixed["allreduce_buckets"] >= 1
    plain = _run_json(cmd, env)
    assert plain["dtype"] == "f32"
    assert mixed["loss"] == mixed["loss"] and abs(mixed["loss"] - plain["loss"]) <= 2e-2 * abs(plain["loss"]), (mixed["loss"], plain["loss"])
    # ... and recorded as a hipGraph (train_step.GraphedTrainStep with a process group: forward, loss -- its all-reduce of the averaging
    # factors included -- and backward are the graph, the bucket all-reduces and AdamW follow each replay): 3 + 1 eager steps, 3 replays
    graphed = _run_json(cmd + ["--dtype", "bf16", "--graph", "--steps", "2", "--warmup", "3"], env)
    assert graphed["graph"] is True and graphed["update_in_graph"] is False and graphed["graph_replays"] == 3 and graphed["optimizer_steps"] == 7
    assert graphed["backend"] == "nccl" and graphed["collectives_active"] is True
    assert graphed["loss"] == graphed["loss"] and abs(graphed["loss"]) < 1e6
    print(f"cfg-4 share, mixed precision: {mixed['ms_per_step']:.1f} ms / step against {plain['ms_per_step']:.1f} ms with f32 products "
          f"(loss {mixed['loss']:.3f} / {plain['loss']:.3f}); recorded as a hipGraph {graphed['ms_per_step']:.1f} ms / step")


def test_cfg5_share_training_step_through_one_rank_rccl():
    """BASELINE configs[4] (SGV3D full config: BSM R101 with the BEV-segmentation branch and the SAM-mask focal supervision,
    exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:295-335, on 8 MI355X): its per-GPU share at FULL size --
    864x1536 images, stride-8 frustum with D = 180, 87-channel BEV map, batch 2 per rank -- one whole data-parallel step
    through a real RCCL group of one rank: detection loss + 500 x semantic focal loss, backward through the fused
    lift-splat adjoint, bucket all-reduces launched inside backward, fused AdamW.  (this share had never run
    in the GPU suite.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "train_bench.py"), "--config", "cfg5", "--batch", "2", "--steps", "2",
           "--warmup", "1", "--dtype", "bf16"]          # the dtype BASELINE configs[4] names: mixed-precision products, f32 master weights
    base = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist = _run_json(cmd, dict(base, SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535"), timeout=1200)
    assert dist["config"] == "cfg5" and dist["backend"] == "nccl" and dist["world_size"] == 1 and dist["collectives_active"] is True
    assert dist["dtype"] == "bf16"
    assert dist["batch_per_gpu"] == 2 and dist["parameters"] > 90e6                      # ResNet-101 + MSCThead + BEV head
    assert dist["allreduces_launched_inside_backward"] == dist["allreduce_buckets"] >= 1
    assert dist["loss"] == dist["loss"] and 0.0 < dist["loss"] < 1e6                     # finite
    f32 = _run_json(cmd[:-2], dict(base, SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537"), timeout=1200)
    assert f32["dtype"] == "f32" and abs(dist["loss"] - f32["loss"]) <= 3e-2 * abs(f32["loss"]), (dist["loss"], f32["loss"])
    print(f"cfg-5 share through 1-rank RCCL, mixed precision: {dist['ms_per_step']:.1f} ms / step at batch 2 (f32 products: {f32['ms_per_step']:.1f} ms, "
          f"loss {dist['loss']:.3f} / {f32['loss']:.3f}), "
          f"{dist['allreduce_bytes_per_step'] / 1e6:.0f} MB all-reduced in {dist['allreduce_buckets']} buckets, "
          f"peak memory {dist['peak_mem_gb']:.1f} GB, loss {dist['loss']:.3f}")


def test_bench_through_one_rank_rccl():
    """bench.py's N>1 protocol (RCCL barrier, MAX all-reduce of the elapsed time, all_gather_object of the per-rank
    records, final barrier before teardown) on a 1-rank RCCL group."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", HSA_ENABLE_IPC_MODE_LEGACY="0")
    rec = _run_json([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                     "--no-roofline"], env)
    assert rec["config"]["backend"] == "nccl (RCCL)" and rec["config"]["world_size"] == 1 and rec["n_gpus"] == 1
    assert rec["value"] > 10.0 and len(rec["config"]["per_rank"]) == 1


def test_overlap_hooks_are_rearmed_by_step_itself():
    """A loop that zeroes gradients with ``model.zero_grad()`` (not the optimiser's) on the no-collectives path never goes
    through zero_grad() / all_reduce_grads() of DataParallelAdamW: step() itself re-arms the per-bucket counters, so the
    next backward is not mistaken for a second backward of the same step."""
    from sgv3d_amd.train_step import DataParallelAdamW
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)).cuda()
    opt = DataParallelAdamW(net.parameters(), lr=1e-3, bucket_bytes=64).overlap_with_backward()
    x = torch.randn(4, 6, device="cuda")
    for _ in range(3):
        net.zero_grad(set_to_none=False)
        net(x).sum().backward()
        opt.step()
    assert opt.steps == 3 and opt._learned and not opt._unused
    assert opt._left == [len(e) for _, _, e in opt.flat.buckets]
