"""Per-calibration state of the view transform: voxel indices + voxel-pooling plan.

``geom_xyz`` (layers/backbones/lss_fpn.py:372-401,487-488 of the reference) and everything derived from it
depend only on the calibration tensors of ``mats_dict`` -- static for a roadside camera -- while the
reference recomputes both on every frame.  ``CalibrationCache`` keeps the int32 index tensor and the CSR
plan of the calibration last seen and tells the backbone whether it may skip the geometry kernel and the
plan build altogether:

* **host fast path**: the calibration tensors handed in are the very same tensor objects, at the same
  versions, as last time (static input buffers of a hipGraph replay, a benchmark loop, a harness that
  keeps its ``mats`` on the device) -> nothing is launched.  The cache holds references to those tensors,
  so their addresses cannot be recycled for other data while it trusts them;
* **device path** (new tensor objects, e.g. the reference harness's ``mats[k].cuda()`` per step): the
  geometry kernel rewrites ``geom`` in place and the cached plan build compares it with the copy the plan
  was built for ON THE DEVICE (no host sync) and rebuilds only on a difference
  (``VoxelPlan(cached=True)``, csrc/voxel_pooling.hip).

Writes through raw pointers (other libraries' kernels) do not bump tensor versions; the calibration tensors
are only ever written by torch ops here (``copy_``), which do.
"""


class CalibrationCache:
    def __init__(self):
        self._src = None          # [(tensor, version)] the cached geometry was computed from
        self._tag = None          # (sweep index, shapes ...) part of the key that is not a tensor
        self.geom = None          # int32 [B, num_cams, D, fH, fW, 3]
        self.plan = None          # VoxelPlan(cached=True) for geom
        self.hits = 0             # forwards that launched neither geometry nor plan kernels
        self.refreshes = 0        # forwards that re-ran the geometry kernel (+ device-side plan check)

    def matches(self, tensors, tag):
        if self._src is None or self._tag != tag or len(tensors) != len(self._src):
            return False
        for t, (old, ver) in zip(tensors, self._src):
            if t is not old or (t is not None and t._version != ver):
                return False
        return True

    def remember(self, tensors, tag):
        self._src = [(t, None if t is None else t._version) for t in tensors]
        self._tag = tag

    def invalidate(self):
        self._src = None
