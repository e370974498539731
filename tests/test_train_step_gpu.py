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
    opt = DataParallelAdamW(model.parameters(), lr=2e-4)

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
