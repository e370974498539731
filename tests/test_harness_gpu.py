"""The drop-in as the UNCHANGED reference harness uses it: ``self.model(sweep_imgs, mats)`` per frame, nothing else
(exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:242-258).  ``BEVHeight.forward`` answers repeated calls of
one signature with a hipGraph replay; these tests hold that replay to the eager forward bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(depth=18, seed=0, bsm=False):
    from sgv3d_amd import synthetic as S
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = (S.small_bsm_conf if bsm else S.small_conf)(depth=depth)
    torch.manual_seed(seed)
    m = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(m, 1)
    return m.cuda(), bc, hc


def _flat(preds):
    return [(t, k, v) for t, task in enumerate(preds) for k, v in sorted(task[0].items())]


def _assert_same(a, b):
    for (t, k, va), (_, _, vb) in zip(_flat(a), _flat(b)):
        assert torch.equal(va, vb), f"task {t} map {k} differs"


@pytest.mark.parametrize("bsm", [False, True])
def test_graph_backed_forward_is_bitwise_the_eager_one(hip, bsm):
    from sgv3d_amd import synthetic as S
    from sgv3d_amd.pipeline import GraphedForward
    model, bc, _ = _model(bsm=bsm)
    scale = bc['final_dim'][0] / 864
    frames = [S.make_images(2, bc['final_dim'], device='cuda', seed=s) for s in range(4)]
    mats = S.make_mats(2, device='cuda', scale=scale)
    with torch.no_grad():
        model.graph_forward = False
        want = [model(f, mats) for f in frames]
        assert not model._graphs
        model.graph_forward = True
        got = []
        for f in frames:
            # fresh calibration tensor objects with every frame, as the harness's `mats[k] = v.cuda()` makes them
            got.append(model(f, {k: v.clone() for k, v in mats.items()}))
    torch.cuda.synchronize()
    (entry,) = model._graphs.values()
    assert isinstance(entry[1], GraphedForward) and entry[1].replays == 3      # call 1 eager, calls 2-4 replayed
    for w, g in zip(want, got):
        _assert_same(w, g)
    # the caller owns what it gets: frame 2's maps are not the static buffers frame 3 / 4 were replayed into
    bases = {(v._base if v._base is not None else v).data_ptr() for preds in got for _, _, v in _flat(preds)}
    assert len(bases) == len(got)
    # the 36 maps of one call are still channel slices of ONE buffer (what get_bboxes' stride asserts rely on)
    assert len({(v._base if v._base is not None else v).data_ptr() for _, _, v in _flat(got[2])}) == 1


def test_graph_backed_forward_follows_calibration_and_weights(hip):
    """Another calibration through the same graph (the plan lives outside the graph and is refreshed eagerly), an in-place
    edit of the calibration tensors, and a weight change (the graph is dropped with the packed weights)."""
    from sgv3d_amd import synthetic as S
    model, bc, _ = _model(seed=3)
    scale = bc['final_dim'][0] / 864
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=5)
    m0 = S.make_mats(1, device='cuda', scale=scale)
    # the second sample of a varied pair is another camera (pitch 12.5 deg, 5.9 m, yaw 1 deg): use it as sample 0's new calibration
    m1 = {k: v[1:2].clone() for k, v in S.make_mats(2, device='cuda', scale=scale).items()}
    with torch.no_grad():
        model.graph_forward = False
        want0, want1 = model(img, m0), model(img, m1)
        g0 = model.backbone.calibration(m0, 0)[0].clone()
        g1 = model.backbone.calibration(m1, 0)[0].clone()
        assert not torch.equal(g0, g1), "the second calibration must move voxel indices for this test to mean anything"
        model.graph_forward = True
        model(img, m0)                                      # eager (first sight)
        _assert_same(want0, model(img, m0))                 # replay
        _assert_same(want1, model(img, m1))                 # replay after the plan refresh
        _assert_same(want0, model(img, m0))                 # and back
        live = {k: v.clone() for k, v in m0.items()}
        _assert_same(want0, model(img, live))
        for k in live:
            live[k].copy_(m1[k])                            # in place: same objects, bumped versions
        _assert_same(want1, model(img, live))
        (entry,) = model._graphs.values()
        assert entry[1].replays == 5
        # weights change -> packed weights and graphs are dropped, the next calls are eager + a new capture
        with torch.no_grad():
            model.head.shared_conv.conv.weight.mul_(0.5)
        model.graph_forward = False
        want2 = model(img, m0)
        model.graph_forward = True
        assert not model._graphs
        model(img, m0)
        _assert_same(want2, model(img, m0))
    torch.cuda.synchronize()


def test_eager_always_where_a_replay_would_be_wrong(hip):
    """Inside someone else's capture, under pipeline.eager_forward, with the instrumented pass on, or with the switch off,
    forward launches its kernels directly; flipping a kernel-selection switch of hip_ops starts a new signature."""
    from sgv3d_amd import hip_ops, synthetic as S
    from sgv3d_amd.pipeline import eager_forward
    model, bc, _ = _model(seed=4)
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=1)
    mats = S.make_mats(1, device='cuda', scale=bc['final_dim'][0] / 864)
    with torch.no_grad():
        with eager_forward(model):
            for _ in range(3):
                model(img, mats)
        assert not model._graphs
        hip_ops.PROFILE = []
        try:
            for _ in range(3):
                model(img, mats)
            assert hip_ops.PROFILE and not model._graphs
        finally:
            hip_ops.PROFILE = None
        model(img, mats); model(img, mats)
        assert len(model._graphs) == 1
        saved = hip_ops.SPLIT_K
        hip_ops.SPLIT_K = not saved
        try:
            model(img, mats)
            assert len(model._graphs) == 2 and list(model._graphs.values())[-1][1] is None      # first sight: eager
        finally:
            hip_ops.SPLIT_K = saved
    model.train()
    model.eval()                                # train() / eval() drop the packed weights and the graphs with them
    assert not model._graphs
    torch.cuda.synchronize()


def test_reference_eval_step_through_the_drop_in(hip):
    """The reference harness's eval_step, restated in sgv3d_amd/harness.py, on host-side calibration tensors: boxes /
    scores / labels as numpy arrays per sample, identical between the eager forward and the graph replays, and equal to
    what decode gives on the eager maps directly."""
    from sgv3d_amd import harness, synthetic as S
    model, bc, _ = _model(seed=6)
    scale = bc['final_dim'][0] / 864
    host = S.make_mats(2, device='cpu', scale=scale)
    frames = [S.make_images(2, bc['final_dim'], device='cuda', seed=10 + s) for s in range(3)]
    with torch.no_grad():
        model.graph_forward = False
        want = [harness.eval_step(model, harness.make_batch(f, host)) for f in frames]
        model.graph_forward = True
        got = [harness.eval_step(model, harness.make_batch(f, host)) for f in frames]
        got += [harness.eval_step(model, harness.make_batch(frames[0].cpu(), host))]          # images from the host, too
    want.append(want[0])
    assert list(model._graphs.values())[0][1].replays == 3
    for w, g in zip(want, got):
        assert len(w) == len(g) == 2
        for (wb, ws, wl, wm), (gb, gs, gl, gm) in zip(w, g):
            assert isinstance(gb, np.ndarray) and gb.ndim == 2 and gb.shape[1] == 9 and gs.shape == gl.shape == (gb.shape[0],)
            assert np.array_equal(wb, gb) and np.array_equal(ws, gs) and np.array_equal(wl, gl) and wm == gm


def test_graph_backed_forward_with_two_sweeps(hip):
    """num_sweeps = 2 through the whole model (the BEV head's trunk takes 2 x 80 channels): the graph's calibration branch
    refreshes one cache entry per sweep, and a replay after BOTH sweeps' cameras moved equals the eager forward."""
    from sgv3d_amd import synthetic as S
    from sgv3d_amd.models.bev_height import BEVHeight
    bc, hc = S.small_conf(depth=18)
    hc['bev_backbone_conf'] = dict(hc['bev_backbone_conf'], in_channels=2 * bc['output_channels'])
    hc['bev_neck_conf'] = dict(hc['bev_neck_conf'], in_channels=[2 * bc['output_channels']] + list(hc['bev_neck_conf']['in_channels'][1:]))
    torch.manual_seed(2)
    model = BEVHeight(bc, hc).eval()
    S.randomize_norm_stats_(model, 1)
    model = model.cuda()
    scale = bc['final_dim'][0] / 864

    def mats_for(shift):
        a, b = S.make_mats(1, device='cuda', scale=scale), S.make_mats(1, device='cuda', scale=scale)
        b['sensor2ego_mats'][:, :, :, 2, 3] += 0.35 + shift
        b['reference_heights'] += 0.35 + shift
        a['sensor2ego_mats'][:, :, :, 2, 3] += shift
        a['reference_heights'] += shift
        return {k: (a[k] if k == 'bda_mat' else torch.cat([a[k], b[k]], 1)) for k in a}
    imgs = torch.cat([S.make_images(1, bc['final_dim'], device='cuda', seed=31), S.make_images(1, bc['final_dim'], device='cuda', seed=32)], 1)
    m0, m1 = mats_for(0.0), mats_for(0.6)
    with torch.no_grad():
        model.graph_forward = False
        want0, want1 = model(imgs, m0), model(imgs, m1)
        assert not torch.equal(want0[0][0]['heatmap'], want1[0][0]['heatmap'])
        model.graph_forward = True
        model(imgs, m0)
        _assert_same(want0, model(imgs, m0))
        _assert_same(want1, model(imgs, m1))
        _assert_same(want0, model(imgs, {k: v.clone() for k, v in m0.items()}))
    torch.cuda.synchronize()
    (entry,) = model._graphs.values()
    assert entry[1].replays == 3


def test_auto_mode_keeps_the_faster_of_replay_and_eager(hip):
    """graph_forward = "auto" (the default): the second call of a signature builds the graph and times it against the eager
    forward on that input; whichever is kept, the outputs are the eager ones bit for bit and the decision is on record."""
    from sgv3d_amd import synthetic as S
    model, bc, _ = _model(seed=8)
    assert model.graph_forward == "auto"
    scale = bc['final_dim'][0] / 864
    frames = [S.make_images(1, bc['final_dim'], device='cuda', seed=40 + s) for s in range(4)]
    mats = S.make_mats(1, device='cuda', scale=scale)
    with torch.no_grad():
        model.graph_forward = False
        want = [model(f, mats) for f in frames]
        model.graph_forward = "auto"
        got = [model(f, {k: v.clone() for k, v in mats.items()}) for f in frames]
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        _assert_same(w, g)
    (entry,) = model._graphs.values()
    assert len(entry) == 3 and set(entry[2]) == {"replay_ms", "eager_ms", "replay_chosen"}
    assert entry[2]["replay_ms"] > 0 and entry[2]["eager_ms"] > 0 and bool(entry[1]) == entry[2]["replay_chosen"]


_FORK_WORKER = """
import json, os, sys
sys.path.insert(0, %r)
import torch
from sgv3d_amd import hip_ops, pipeline, synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight
bsm, fork_refresh = bool(int(sys.argv[1])), bool(int(sys.argv[2]))
bc, hc = (S.small_bsm_conf if bsm else S.small_conf)(depth=18)
torch.manual_seed(11)
model = BEVHeight(bc, hc).eval()
S.randomize_norm_stats_(model, 1)
model = model.cuda()
frames = [S.make_images(1, bc['final_dim'], device='cuda', seed=50 + s) for s in range(3)]
mats = S.make_mats(1, device='cuda', scale=bc['final_dim'][0] / 864)
flat = lambda preds: [v for task in preds for _, v in sorted(task[0].items())]
with torch.no_grad():
    model.graph_forward = False
    want = [flat(model(f, mats)) for f in frames]
    hip_ops.PARALLEL_BRANCHES = True                 # (part of hip_ops.switch_state: a signature of its own)
    pipeline.FORK_REFRESH = fork_refresh
    model.graph_forward = True
    got = [flat(model(f, {k: v.clone() for k, v in mats.items()})) for f in frames]
torch.cuda.synchronize()
same = all(torch.equal(a, b) for w, g in zip(want, got) for a, b in zip(w, g))
print(json.dumps({"replays": list(model._graphs.values())[-1][1].replays, "same": same}))
"""


@pytest.mark.parametrize("bsm,fork_refresh", [(False, False), (True, False), (False, True)])
def test_parallel_graph_branches_are_bitwise_the_sequential_forward(hip, bsm, fork_refresh, tmp_path):
    """hip_ops.run_parallel (SGV3D_PARALLEL_BRANCHES=1; measured slower, off by default -- DESIGN 3.6) and the forked calibration
    refresh of the graphed forward (SGV3D_GRAPH_FORK_REFRESH=1; in line by default since round 6): shortcut convolutions, SECONDFPN
    levels, ASPP pooled branch, gate MLPs / context branch, MSCThead scales and tasks, the refresh as forked branches of the
    captured graph give the bytes of the sequential forward.  In a process of its own: hipGraphLaunch of graphs with parallel
    branches crashed inside ROCm 7.2 (hip::Graph::UpdateStreams) once a few dozen graphs had been created in one process --
    which is why no default path replays such a graph any more."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fork_worker.py"
    script.write_text(_FORK_WORKER % root)
    r = subprocess.run([sys.executable, str(script), str(int(bsm)), str(int(fork_refresh))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec == {"replays": 2, "same": True}, rec


def test_eval_step_under_inference_mode(hip):
    """Recent Lightning releases run validation under ``torch.inference_mode()``: tensors made there have no version counter
    (the calibration cache and the decode reuse key on versions).  Same boxes as under ``no_grad``; nothing is cached on the
    strength of a version that does not exist."""
    from sgv3d_amd import harness, synthetic as S
    model, bc, _ = _model(seed=9)
    host = S.make_mats(1, device='cpu', scale=bc['final_dim'][0] / 864)
    frames = [S.make_images(1, bc['final_dim'], device='cuda', seed=60 + s) for s in range(3)]
    with torch.no_grad():
        model.graph_forward = False
        want = [harness.eval_step(model, harness.make_batch(f, host)) for f in frames]
    for mode in (False, True):
        model.graph_forward = mode
        model._graphs = {}
        with torch.inference_mode():
            got = [harness.eval_step(model, harness.make_batch(f, host)) for f in frames]
        for w, g in zip(want, got):
            for (wb, ws, wl, _), (gb, gs, gl, _) in zip(w, g):
                assert np.array_equal(wb, gb) and np.array_equal(ws, gs) and np.array_equal(wl, gl)


def test_graph_captured_under_inference_mode_replays_under_no_grad(hip):
    """Lightning's validation loop runs under torch.inference_mode(); a graph captured there used to hold inference
    tensors as its static inputs, and the first replay under plain torch.no_grad() (user code after trainer.validate()) raised
    'Inplace update to inference tensor outside InferenceMode'.  Both orders, bitwise the eager forward."""
    from sgv3d_amd import synthetic as S
    model, bc, _ = _model(seed=6)
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=2)
    img2 = S.make_images(1, bc['final_dim'], device='cuda', seed=3)
    mats = S.make_mats(1, device='cuda', scale=bc['final_dim'][0] / 864)
    model.graph_forward = False
    with torch.no_grad():
        want, want2 = model(img, mats), model(img2, mats)
    model.graph_forward = True
    with torch.inference_mode():
        model(img, mats)                                    # first sight: eager
        got = model(img, mats)                              # capture + replay, under inference mode
        (entry,) = model._graphs.values()
        assert entry[1] and entry[1].replays == 1 and not entry[1].in_imgs.is_inference()
    _assert_same(want, got)
    with torch.no_grad():
        _assert_same(want2, model(img2, mats))              # the same graph, replayed under no_grad: in-place input copies
        assert len(model._graphs) == 1 and entry[1].replays == 2
    with torch.inference_mode():
        _assert_same(want, model(img, mats))
    # a NEW signature captured under no_grad while the graph captured under inference mode is alive: torch fills the RNG generator's
    # graph-state tensors in place at every capture_begin, and they were allocated by that first capture (pipeline.capture_begin)
    import warnings
    img_b2 = torch.cat([img, img2])
    mats_b2 = S.make_mats(2, device='cuda', scale=bc['final_dim'][0] / 864)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        model(img_b2, mats_b2)
        model(img_b2, mats_b2)
    assert len(model._graphs) == 2 and all(e[1] and e[1].replays >= 1 for e in model._graphs.values())
    torch.randn(4, device='cuda')                            # (a failed capture_begin used to leave the generator "capturing")
    torch.cuda.synchronize()


def test_decode_config_is_part_of_the_graph_signature(hip):
    """The graph bakes bbox_coder / test_cfg into the decode kernels' arguments: a changed score threshold must not be answered
    by the old graph's decode."""
    from sgv3d_amd import synthetic as S
    model, bc, _ = _model(seed=7)
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=4)
    mats = S.make_mats(1, device='cuda', scale=bc['final_dim'][0] / 864)
    with torch.no_grad():
        for t in model.head.task_heads:                     # spread the heatmap logits so that detections exist
            t.heatmap[1].weight.mul_(40.0)
            t.heatmap[1].bias.fill_(-1.0)
        model.graph_forward = True
        model(img, mats)
        n_low = sum(len(s) for _, s, _ in model.get_bboxes(model(img, mats)))
        assert len(model._graphs) == 1
        model.head.bbox_coder_cfg = dict(model.head.bbox_coder_cfg, score_threshold=0.6)
        model(img, mats)                                    # first sight of the new signature: eager
        assert len(model._graphs) == 2
        n_high_graph = sum(len(s) for _, s, _ in model.get_bboxes(model(img, mats)))
        model.graph_forward = False
        n_high_eager = sum(len(s) for _, s, _ in model.get_bboxes(model(img, mats)))
    assert n_high_graph == n_high_eager and n_high_graph < n_low, (n_low, n_high_graph, n_high_eager)


def test_camera_gates_are_cached_per_calibration_and_equal_the_per_frame_ones(hip):
    """The height net's SE gates depend on the 27 calibration numbers only (lss_fpn.py:208-246): kept in the calibration entry,
    rewritten in place when the calibration changes, bitwise the gates computed per frame."""
    from sgv3d_amd import hip_ops, synthetic as S
    model, bc, _ = _model(seed=8)
    scale = bc['final_dim'][0] / 864
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=5)
    m0 = S.make_mats(1, device='cuda', scale=scale)
    m1 = {k: v[1:2].clone() for k, v in S.make_mats(2, device='cuda', scale=scale).items()}
    model.graph_forward = False
    bb = model.backbone
    with torch.no_grad():
        for m in (m0, m1, m0):
            hip_ops.PROFILE = []
            out = model(img, m)
            first = [r[0] for r in hip_ops.PROFILE]
            hip_ops.PROFILE = []
            again = model(img, m)
            second = [r[0] for r in hip_ops.PROFILE]
            hip_ops.PROFILE = None
            assert first.count("dense") >= 10 and second.count("dense") <= 2     # the 8 gate launches ran once per calibration (2 left: ASPP pooled branch)
            _assert_same(out, again)
            fresh = bb.height_net.camera_gates(m, img.device)
            for a, b in zip(bb.calib_cache.entry(0).gates, fresh):
                assert torch.equal(a, b)
    torch.cuda.synchronize()


def test_cached_camera_gates_follow_a_weight_refresh_on_a_static_camera(hip):
    """A static-camera stream (the same calibration tensors every frame) across weight changes: ``refresh()`` drops the packed state
    the cached gates were computed from, and the calibration entry is keyed on the state's generation number -- the ``id()`` of the
    freed state dict, the earlier key, is handed out again by CPython more often than not, and the old weights' gates were then
    reused silently.  After every in-place change of the gate MLP's weights the cached gates equal freshly computed ones."""
    from sgv3d_amd import synthetic as S
    model, bc, _ = _model(seed=9)
    scale = bc['final_dim'][0] / 864
    img = S.make_images(1, bc['final_dim'], device='cuda', seed=6)
    mats = S.make_mats(1, device='cuda', scale=scale)
    model.graph_forward = False
    bb = model.backbone
    gens = []
    with torch.no_grad():
        model(img, mats)
        for it in range(6):
            bb.height_net.context_mlp.fc1.weight.mul_(1.5)            # (version bump -> stamp -> refresh on the next forward)
            bb.height_net.height_mlp.fc2.bias.add_(0.3)
            model(img, mats)
            gens.append(bb.height_net._hip_gen)
            fresh = bb.height_net.camera_gates(mats, img.device)
            for a, b in zip(bb.calib_cache.entry(0).gates, fresh):
                assert torch.equal(a, b), it
    assert gens == sorted(set(gens)) and len(gens) == 6                # a new, larger generation per refresh
    torch.cuda.synchronize()


def test_aspp_pooled_branch_folded_into_the_bias_matches_the_concat_form(hip):
    """Batch 1, f32: ASPP's pooled branch is one vector per image, so its share of conv1 is a per-image bias (2048 instead of 2560
    input channels, no broadcast launch).  Against the five-branch concat form: same function, f32 rounding apart."""
    import sgv3d_amd.layers.backbones.lss_fpn as L
    torch.manual_seed(3)
    aspp = L.ASPP(128, 128).eval()
    with torch.no_grad():
        for m in aspp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0); m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
    aspp = aspp.cuda()
    x = torch.randn(1, 20, 28, 128, device='cuda')
    saved = L.FOLD_ASPP_POOL
    try:
        with torch.no_grad():
            L.FOLD_ASPP_POOL = True
            folded = aspp.hip_forward(x)
            L.FOLD_ASPP_POOL = False
            plain = aspp.hip_forward(x)
            both = aspp.hip_forward(torch.cat([x, x]))      # batch 2: always the concat form
    finally:
        L.FOLD_ASPP_POOL = saved
    scale = float(plain.abs().max())
    assert float((folded - plain).abs().max()) <= 1e-5 * scale
    assert torch.equal(both[0], plain[0]) or float((both[0] - plain[0]).abs().max()) <= 1e-5 * scale
