"""Frames-in-flight execution of the camera->BEV forward: hipGraphs on several HIP streams.

A batch-1 forward is a chain of ~130 kernels, many of which cannot fill 256 CUs on their own
(ResNet layer 3/4, the 32x32 / 64x64 BEV trunk stages, the SECONDFPN levels, every tiny
HeightNet gate kernel).  ``FramePipeline`` captures the whole forward into one hipGraph per slot
(each slot has its own stream and its own activation pool; weights are shared) and replays
consecutive frames round-robin over the slots, so the kernels of frame i+1 occupy the CUs that the
narrow layers of frame i leave idle.  Frames stay independent batch-1 forwards; nothing is batched
or skipped.

Calibration (geometry + voxel-pooling plan) is per slot and lives OUTSIDE the captured graph: the graph
reads the slot's plan buffer, and ``submit`` refreshes that buffer eagerly on the slot's stream only when it
is handed calibration tensors other than the ones the slot already holds (sgv3d_amd/calibration.py; the
refresh itself re-runs the plan build only if the voxel indices really changed, decided on the device).
A roadside camera's calibration is static, so in steady state a frame costs no geometry or plan kernels.

Usage (static input buffers, as for any graph replay)::

    pipe = FramePipeline(model, imgs, mats, slots=3)
    for frame in frames:
        slot = pipe.submit(frame_imgs, frame_mats)   # copies into the slot's static inputs, replays its graph
        ...
        preds = pipe.result(slot)                    # orders the caller's stream after that slot's frame

Stream contract: ``submit`` orders the slot's stream after everything already enqueued on the caller's current
stream (producers of ``imgs`` / ``mats``, consumers of the slot's previous outputs); ``result`` makes the caller's
current stream wait for the frame and returns the slot's static output tensors, valid until that slot is submitted
again (``slots`` submits later) -- clone what must live longer.
"""
import os
import warnings

import torch

from . import hip_ops
from .calibration import CalibrationCache, tensor_version

# what a failed stream capture raises (HIP errors surface as RuntimeError / torch.AcceleratorError, a library call that
# refuses to run under capture as SGV3DError); anything else is a bug and propagates
from ._lib import SGV3DError  # noqa: E402
CAPTURE_ERRORS = (RuntimeError, SGV3DError)
# GraphedForward: the calibration refresh as a forked branch of the graph (under the image backbone) instead of in line
FORK_REFRESH = os.environ.get("SGV3D_GRAPH_FORK_REFRESH", "0") == "1"


class eager_forward:
    """``with eager_forward(model):`` -- ``model(...)`` launches its kernels directly inside the block instead of replaying
    the hipGraph ``BEVHeight.forward`` keeps per input signature (used where the caller captures or instruments the
    launches itself)."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.saved = getattr(self.model, "_graph_suspended", 0)
        self.model._graph_suspended = self.saved + 1
        return self

    def __exit__(self, *exc):
        self.model._graph_suspended = self.saved
        return False


def static_copy(t):
    """A clone of ``t`` that is an ordinary tensor whatever mode the caller is in.  Static graph inputs are written in place on
    every replay; a clone made under ``torch.inference_mode()`` (Lightning's validation loop) would be an inference tensor, and
    the first replay under plain ``torch.no_grad()`` would raise 'Inplace update to inference tensor outside InferenceMode'."""
    with torch.inference_mode(False):
        return t.clone()


def capture_begin(graph, pool=None, error_mode=None):
    """``graph.capture_begin(pool=pool)`` outside inference mode.  The first live graph of a process makes torch allocate the
    RNG generator's two graph-state tensors and EVERY later ``capture_begin`` fills them in place; allocated by a capture under
    ``torch.inference_mode()`` (Lightning's validation loop) they are inference tensors, and the next capture under plain
    ``torch.no_grad()`` -- a new input signature after ``trainer.validate()`` -- raises 'Inplace update to inference tensor outside
    InferenceMode' from inside ``capture_begin``, which also leaves the generator marked as capturing (every later
    ``torch.randn(device='cuda')`` of the process then fails).  Ordinary tensors can be filled from either mode.
    ``error_mode``: torch's ``capture_error_mode`` ("global" by default; "thread_local" lets OTHER threads -- the process group's
    watchdog polling its events -- keep making calls a capture forbids)."""
    kw = {} if pool is None else {"pool": pool}
    if error_mode is not None:
        kw["capture_error_mode"] = error_mode
    with torch.inference_mode(False):
        graph.capture_begin(**kw)


def _clone_aliased(obj, memo):
    """Deep copy of a nest of tensors that keeps their aliasing: tensors that share a storage become views of ONE copy of it
    (keyed by the storage, not by ``_base``: views made under ``torch.inference_mode()`` do not record their base)."""
    if torch.is_tensor(obj):
        st = obj.untyped_storage()
        c = memo.get(st.data_ptr())
        if c is None:
            whole = torch.empty(0, dtype=obj.dtype, device=obj.device).set_(st)      # the storage as one flat tensor
            c = memo[st.data_ptr()] = whole.clone()
        return c.as_strided(obj.size(), obj.stride(), obj.storage_offset())
    if isinstance(obj, dict):
        return {k: _clone_aliased(v, memo) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_clone_aliased(v, memo) for v in obj)
    return obj


class GraphedForward:
    """The inference forward of one input signature as ONE hipGraph replay on the caller's stream, behind
    ``BEVHeight.forward`` (so a harness that only ever calls ``model(imgs, mats)`` -- the reference's eval_step,
    exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:242-258 -- gets graph speed without knowing this class).

    What the graph holds, besides the forward itself:
    * the **calibration refresh** (the device-side "same numbers as last frame?" check, then calib_prep + geometry kernel + the
      device-gated plan rebuild + the camera gates, all of which return at once for a static camera): the harness hands fresh
      calibration tensors with every frame.  In line in front of the forward since round 6 (``SGV3D_GRAPH_FORK_REFRESH=1``: as a
      forked branch under the image backbone, joined right in front of the lift-splat gather -- rounds 4-5; graphs with parallel
      branches turned out to crash ROCm 7.2's hipGraphLaunch in long-lived processes).  The plan is rebuilt only when the voxel
      indices really changed (decided on the device);
    * the **box decode** of the forward's own output (``BEVHeightHead.decode_device``: top-K, box assembly, circle NMS, task
      merge), right behind the head: ``BEVHeight.get_bboxes`` on the very maps this call returned finds them decoded and only
      reads the detection counts back.

    Static input buffers (``imgs`` and the calibration tensors are copied in on the caller's stream, the latter as one
    multi-tensor copy); outputs are COPIED out of the graph's static buffers (one 18 MB device copy at cfg-2 plus the 0.1 MB
    of decoded boxes): what the caller receives is its own, like the result of an eager call.  Same kernels on the same
    buffers in the same order: bitwise the eager forward (tests/test_harness_gpu.py)."""

    def __init__(self, model, imgs, mats):
        # (no reference to the model is kept: it owns this object, and a cycle would hold the graph's activation pool
        # until the cycle collector runs)
        dev = imgs.device
        self.in_imgs = static_copy(imgs)
        self.in_mats = {k: static_copy(v) for k, v in mats.items()}
        self._mat_keys = sorted(self.in_mats)
        self._mat_dst = [self.in_mats[k] for k in self._mat_keys]
        self.cache = CalibrationCache()
        self.replays = 0
        self.decoded = None
        # (streams that are neither each other, nor the caller's, nor a branch stream of hip_ops.run_parallel: torch hands its 32
        # pool streams out round-robin)
        side = hip_ops.distinct_stream(dev)
        branch = hip_ops.distinct_stream(dev, (side,))
        self._streams = (side, branch)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        own = model.backbone.calib_cache
        model.backbone.calib_cache = self.cache
        sweeps = range(int(self.in_imgs.shape[1]))
        try:
            with torch.cuda.stream(side), torch.no_grad(), eager_forward(model):
                preds = model(self.in_imgs, self.in_mats)  # this signature's geometry + plan, eagerly
                model.head.decode_device(preds)            # (first call of the decode kernels outside any capture)
                del preds
                side.synchronize()
                # TWO graphs that share one memory pool: a short one (image stem + first backbone stage, ~15 launches, 0.5 ms
                # of GPU work) and the rest (~160 launches).  Submitting a long hipGraph takes the host a few hundred
                # microseconds BEFORE its first kernel starts; with one frame in flight -- the harness waits for every
                # frame's boxes -- the GPU idles through that.  The short graph starts at once and the long one is submitted
                # under it (SGV3D_GRAPH_SPLIT=0: one graph).
                self.graphs = [torch.cuda.CUDAGraph()]
                pool = torch.cuda.graph_pool_handle()
                split = os.environ.get("SGV3D_GRAPH_SPLIT", "1") != "0"

                def fork_refresh():
                    if not FORK_REFRESH:
                        # in line, in front of the forward.  With the device-side "same numbers as last frame?" check
                        # (calibration.py, round 6) the refresh of a static camera is ~15 launches that return at once (~50 us),
                        # and the graph stays a LINEAR chain: hipGraphLaunch of graphs with parallel branches crashed inside
                        # hip::Graph::UpdateStreams (ROCm 7.2) after a few dozen such graphs had been created in one process --
                        # order-dependent, seen in the GPU suite in round 6.
                        self.cache.invalidate()
                        for sweep in sweeps:
                            model.backbone.calibration(self.in_mats, sweep)
                        return
                    branch.wait_stream(side)               # SGV3D_GRAPH_FORK_REFRESH=1: the refresh recorded on its own branch ...
                    with torch.cuda.stream(branch):
                        self.cache.invalidate()
                        for sweep in sweeps:
                            model.backbone.calibration(self.in_mats, sweep)
                    for sweep in sweeps:
                        self.cache.entry(sweep).join_stream = branch     # ... and joined by its first reader (calibration.py)

                def cut(tag):                              # (hip_ops.graph_split_point, once: sweep 0's image backbone)
                    hip_ops._GRAPH_SPLIT_HOOK = None
                    self.graphs[-1].capture_end()
                    self.graphs.append(torch.cuda.CUDAGraph())
                    capture_begin(self.graphs[-1], pool)
                    fork_refresh()                         # a fork has to rejoin inside the graph it was recorded in: the last one
                torch.cuda.synchronize(dev)
                capture_begin(self.graphs[0], pool)
                try:
                    if split:
                        hip_ops._GRAPH_SPLIT_HOOK = cut
                    else:
                        fork_refresh()
                    self.outputs = model(self.in_imgs, self.in_mats)
                    if hip_ops._GRAPH_SPLIT_HOOK is not None:      # (a backbone without the cut point)
                        hip_ops._GRAPH_SPLIT_HOOK = None
                        fork_refresh()
                    if FORK_REFRESH:
                        side.wait_stream(branch)           # (a join of its own if the forward never asked for the plan)
                    self.decoded = model.head.decode_device(self.outputs)
                finally:
                    hip_ops._GRAPH_SPLIT_HOOK = None
                    self.graphs[-1].capture_end()
        finally:
            for sweep in sweeps:
                self.cache.entry(sweep).join_stream = None
            model.backbone.calib_cache = own
        cur.wait_stream(side)

    def __del__(self):
        for st in getattr(self, '_streams', ()):
            hip_ops.release_stream(st)

    def __call__(self, model, imgs, mats):
        with torch.no_grad():
            self.in_imgs.copy_(imgs, non_blocking=True)
            torch._foreach_copy_(self._mat_dst, [mats[k] for k in self._mat_keys], non_blocking=True)
            for g in self.graphs:
                g.replay()
            self.replays += 1
            out = _clone_aliased(self.outputs, {})
            return out, self.decoded.clone()


class FramePipeline:
    def __init__(self, model, imgs, mats, slots=2, use_graph=True, strict=False):
        assert imgs.is_cuda, "FramePipeline runs on the GPU"
        self.model = model
        self.capture_error = None                  # the exception that made the pipeline fall back to eager launches
        self.device = imgs.device
        self.slots = max(1, int(slots))
        self.streams = []
        for _ in range(self.slots):                # pairwise distinct HIP streams (hip_ops.distinct_stream)
            self.streams.append(hip_ops.distinct_stream(imgs.device, self.streams))
        self.in_imgs = [static_copy(imgs) for _ in range(self.slots)]
        self.in_mats = [{k: static_copy(v) for k, v in mats.items()} for _ in range(self.slots)]
        self.caches = [CalibrationCache() for _ in range(self.slots)]
        self._last_mats = [None] * self.slots      # [(tensor, version)] of the mats last copied into the slot
        self.outputs = [None] * self.slots
        self.graphs = []
        self.done = [torch.cuda.Event() for _ in range(self.slots)]
        self._next = 0
        self._own_cache = model.backbone.calib_cache
        # The first forward times the per-layer candidates (tile, split-K, Winograd variant) as ISOLATED launches (hip_ops.TUNE_STREAMS
        # = 1 unless SGV3D_TUNE_STREAMS says otherwise).  Rounds 3-5 timed them as `slots` concurrent copies; round 6 measured that the
        # chip-filling launches of concurrent frames run one after the other on the device (a kernel trace of three frames in flight:
        # the frame costs the sum of its kernels' isolated durations, the overlap hides the gaps between them) -- the isolated
        # duration is what a frame pays, and the choices made on it are 1-2 % faster with three frames in flight as well.
        with torch.no_grad(), eager_forward(model):
            model(imgs, mats)                      # packs weights / tunes tiles outside any capture
            torch.cuda.synchronize(imgs.device)
            if use_graph:
                try:
                    for i, s in enumerate(self.streams):
                        g = torch.cuda.CUDAGraph()
                        s.wait_stream(torch.cuda.current_stream(imgs.device))
                        with torch.cuda.stream(s), self._slot(i):
                            model(self.in_imgs[i], self.in_mats[i])      # builds the slot's geometry + plan, eagerly
                            torch.cuda.synchronize(imgs.device)
                            # (outside inference mode, see capture_begin; the forward's tensors are then ordinary ones)
                            with torch.inference_mode(False), torch.no_grad(), torch.cuda.graph(g, stream=s):
                                self.outputs[i] = model(self.in_imgs[i], self.in_mats[i])
                        torch.cuda.current_stream(imgs.device).wait_stream(s)
                        self.graphs.append(g)
                except CAPTURE_ERRORS as e:
                    # a forward that cannot be captured (an operator that synchronises, an allocator the capture rejects):
                    # the same kernels run as eager launches on the slot streams -- 20-30 % slower at batch 1, so say so
                    # (strict=True raises instead).  Programming errors (TypeError, AssertionError, ...) are not caught.
                    self.graphs = []
                    self.outputs = [None] * self.slots
                    self.capture_error = e
                    torch.cuda.synchronize(imgs.device)
                    if strict:
                        raise
                    warnings.warn(f"FramePipeline: hipGraph capture failed ({type(e).__name__}: {e}); running eager launches on "
                                  f"the slot streams instead (pass strict=True to raise)", RuntimeWarning, stacklevel=2)
        self.use_graph = bool(self.graphs)

    def __del__(self):
        for st in getattr(self, 'streams', ()):
            hip_ops.release_stream(st)

    class _Slot:
        def __init__(self, pipe, i):
            self.pipe, self.i = pipe, i

        def __enter__(self):
            self.pipe.model.backbone.calib_cache = self.pipe.caches[self.i]

        def __exit__(self, *exc):
            self.pipe.model.backbone.calib_cache = self.pipe._own_cache
            return False

    def _slot(self, i):
        """The model's backbone uses slot i's calibration cache inside this context."""
        return FramePipeline._Slot(self, i)

    def replay(self, slot=None):
        """Run one forward on the next (or given) slot with whatever its static inputs hold."""
        i = self._next if slot is None else slot
        self._next = (i + 1) % self.slots
        with torch.cuda.stream(self.streams[i]), torch.no_grad():
            if self.graphs:
                self.graphs[i].replay()
            else:
                with self._slot(i), eager_forward(self.model):
                    self.outputs[i] = self.model(self.in_imgs[i], self.in_mats[i])
            self.done[i].record()
        return i

    def _same_mats(self, i, mats):
        last = self._last_mats[i]
        if last is None or len(last) != len(mats):
            return False
        return all(k in last and last[k][0] is v and last[k][1] is not None and last[k][1] == tensor_version(v) for k, v in mats.items())

    def submit(self, imgs, mats):
        """Copy a frame into the next slot's static inputs (on that slot's stream) and run it."""
        i = self._next
        s = self.streams[i]
        # the frame may have been produced on the caller's stream (a preprocessing kernel, a non_blocking H2D copy),
        # and kernels reading this slot's previous outputs may still be queued there
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s), torch.no_grad():
            self.in_imgs[i].copy_(imgs, non_blocking=True)
            if imgs.is_cuda:
                imgs.record_stream(s)
            if not self._same_mats(i, mats):
                keys = list(mats)
                torch._foreach_copy_([self.in_mats[i][k] for k in keys], [mats[k] for k in keys], non_blocking=True)   # one launch
                for v in mats.values():
                    if v.is_cuda:
                        v.record_stream(s)
                self._last_mats[i] = {k: (v, tensor_version(v)) for k, v in mats.items()}
                if self.graphs:
                    # the captured graph holds no geometry / plan kernels: bring the slot's plan buffer up to date
                    # here (a no-op on the device when the new calibration yields the same voxel indices).
                    # (Round 6 measured the alternatives: the refresh recorded into a second graph per slot, inline -- the same frame
                    # rate as these eager launches; as a forked branch under the image backbone (GraphedForward's layout of rounds 4-5) -- 10 %
                    # SLOWER with three frames in flight, cached calibration included: the extra streams alone cost it, and more
                    # hardware queues (GPU_MAX_HW_QUEUES 8 / 16) made it worse, 2 queues likewise.)
                    with self._slot(i):
                        self.model.backbone.calibration(self.in_mats[i], 0)
        return self.replay(i)

    def result(self, slot, wait_host=False):
        """Outputs of the frame last submitted to ``slot``.  The caller's current stream is ordered after the frame;
        ``wait_host=True`` additionally blocks the host until it is done (needed before reading from the CPU)."""
        torch.cuda.current_stream(self.device).wait_event(self.done[slot])
        if wait_host:
            self.done[slot].synchronize()
        return self.outputs[slot]
