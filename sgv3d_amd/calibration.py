"""Per-calibration state of the view transform: voxel indices + voxel-pooling plan (+ the height net's camera-aware gates).

``geom_xyz`` (layers/backbones/lss_fpn.py:372-401,487-488 of the reference) and everything derived from it
depend only on the calibration tensors of ``mats_dict`` -- static for a roadside camera -- while the
reference recomputes both on every frame.  ``CalibrationCache`` keeps the int32 index tensor and the CSR
plan of the calibration last seen and tells the backbone whether it may skip the geometry kernel and the
plan build altogether:

* **host fast path**: the calibration tensors handed in are the very same tensor objects, at the same
  versions, as last time (static input buffers of a hipGraph replay, a benchmark loop, a harness that
  keeps its ``mats`` on the device) -> nothing is launched.  The cache holds references to those tensors,
  so their addresses cannot be recycled for other data while it trusts them;
* **device path** (new tensor objects, e.g. the reference harness's ``mats[k].cuda()`` per step): one small kernel
  compares the tensors' numbers with those of the last change ON THE DEVICE (``sgv3d_calib_changed``, round 6); the
  geometry kernels and the gate MLPs take its flag and return at once when nothing changed, otherwise the geometry
  kernel rewrites ``geom`` in place and the cached plan build compares it with the copy the plan was built for -- also
  on the device, no host sync -- and rebuilds only on a difference (``VoxelPlan(cached=True)``, csrc/voxel_pooling.hip).

Writes through raw pointers (other libraries' kernels) do not bump tensor versions; the calibration tensors
are only ever written by torch ops here (``copy_``), which do.
"""


def tensor_version(t):
    """``t._version``; None for a tensor made under ``torch.inference_mode()`` (it has no version counter -- Lightning's
    validation loop runs under inference mode by default in recent releases): such a tensor never matches a remembered one,
    i.e. it is treated as new content every time (geometry kernel + device-side compare, nothing wrong, nothing cached on the
    host)."""
    if t is None:
        return None
    try:
        if t.is_inference():
            return None
    except AttributeError:      # (not a tensor)
        return None
    return t._version


class CalibrationCache:
    """One entry per sweep index (multi-sweep inputs carry one calibration per sweep; with a single slot consecutive
    sweeps would evict each other and every sweep would pay the geometry kernel + plan rebuild on every frame).
    ``geom`` / ``plan`` / ``matches`` / ``remember`` of the cache itself are those of sweep 0, the key frame.

    Stream safety: an entry records the stream it was (re)built on and an event behind the last build kernel.  A host
    fast-path hit on ANOTHER stream makes that stream wait for the event before it reads ``geom`` / ``plan.buf`` (the
    caller cannot know that a hidden plan build is still in flight on the first stream).  ``invalidate()`` is for callers
    that rewrite the calibration tensors out of band (raw pointers, ``.data``): tensor versions do not see such writes."""

    class Entry:
        __slots__ = ("_src", "_tag", "geom", "plan", "gates", "event", "stream", "join_stream", "content", "changed", "gate_tmp")

        def __init__(self):
            self._src = None          # [(tensor, version)] the cached geometry was computed from
            self._tag = None          # (sweep index, shapes ...) part of the key that is not a tensor
            self.geom = None          # int32 [B, num_cams, D, fH, fW, 3]
            self.plan = None          # VoxelPlan(cached=True) for geom
            self.gates = None         # key frame only: the camera-aware SE gate vectors of the height net (a function of the
                                      # calibration alone, lss_fpn.py:208-246), persistent buffers rewritten in place
            self.event = None         # recorded behind the last (re)build
            self.stream = None        # cuda_stream handle of that build
            self.join_stream = None   # stream capture only: the side stream the refresh was recorded on (see join_capture)
            self.content = None       # device copy of the calibration tensors' bytes at the last change (sgv3d_calib_changed)
            self.changed = None       # int32 device flag of the refresh in flight: 0 = same numbers as `content`, kernels skip
            self.gate_tmp = None      # the gate MLPs' intermediate vectors (persistent: the gated launches write in place)

        def matches(self, tensors, tag):
            if self._src is None or self._tag != tag or len(tensors) != len(self._src):
                return False
            for t, (old, ver) in zip(tensors, self._src):
                if t is not old or (t is not None and (ver is None or tensor_version(t) != ver)):
                    return False
            return True

        def remember(self, tensors, tag):
            self._src = [(t, tensor_version(t)) for t in tensors]
            self._tag = tag

        def mark_built(self, device):
            """Call right after enqueueing the geometry / plan kernels on the current stream."""
            import torch
            cur = torch.cuda.current_stream(device)
            if torch.cuda.is_current_stream_capturing():
                return                # inside a capture the graph's own edges order the kernels
            if self.event is None:
                self.event = torch.cuda.Event()
            self.event.record(cur)
            self.stream = cur.cuda_stream

        def join_capture(self, device):
            """Inside a stream capture that recorded this entry's refresh on a forked side stream
            (``pipeline.GraphedForward``: geometry + plan check run as a parallel branch of the graph, under the image
            backbone): the first reader of ``geom`` / ``plan`` makes the capturing stream wait for that branch -- the edge
            that orders the gather behind the plan."""
            if self.join_stream is not None:
                import torch
                torch.cuda.current_stream(device).wait_stream(self.join_stream)
                self.join_stream = None

        def order_after_build(self, device):
            """Host-path hit: make the current stream wait for the build if it ran on another stream."""
            import torch
            if self.event is None or torch.cuda.is_current_stream_capturing():
                return
            cur = torch.cuda.current_stream(device)
            if cur.cuda_stream != self.stream:
                cur.wait_event(self.event)

    def __init__(self):
        self._entries = {0: CalibrationCache.Entry()}
        self.hits = 0             # forwards that launched neither geometry nor plan kernels
        self.refreshes = 0        # forwards that re-ran the geometry kernel (+ device-side plan check)

    def entry(self, sweep_index=0):
        e = self._entries.get(int(sweep_index))
        if e is None:
            e = self._entries[int(sweep_index)] = CalibrationCache.Entry()
        return e

    # sweep 0 (the key frame) through the cache object itself
    @property
    def geom(self):
        return self._entries[0].geom

    @property
    def plan(self):
        return self._entries[0].plan

    def matches(self, tensors, tag):
        return self._entries[0].matches(tensors, tag)

    def remember(self, tensors, tag):
        self._entries[0].remember(tensors, tag)

    def invalidate(self):
        """Forget what the cached geometry was computed from (every sweep): the next forward goes through the device path (the
        calibration tensors' numbers compared with the last ones on the device, geometry kernel and plan check when they
        differ).  Needed after writing calibration tensors through raw pointers / ``.data``."""
        for e in self._entries.values():
            e._src = None
