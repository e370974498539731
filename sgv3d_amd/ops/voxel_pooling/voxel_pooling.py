"""``voxel_pooling(geom_xyz, input_features, voxel_num) -> Tensor[B, C, Y, X]``

Mirror of the reference's autograd operator (ops/voxel_pooling/voxel_pooling.py:10-72): same
callable, argument meaning, asserts, non-differentiable ``geom_xyz``, permuted-view return and
``(None, grad_features, None)`` backward.  What is different underneath:

* the kernel is hand-written HIP for gfx950 behind the C ABI (``include/sgv3d_hip.h``);
* two modes (``set_mode``): ``"planned"`` (default) — deterministic sort-by-voxel + segmented
  gather-reduce, every output row written once with 16-B stores; ``"atomic"`` — the reference's
  float-atomic scatter (order-nondeterministic, voxel_pooling_forward_cuda.cu:30-33);
* the 149 MB ``grad_input_features`` memset and ``pos_memo`` are only produced when a gradient is
  actually required (the reference allocates both unconditionally, voxel_pooling.py:29,40);
* ``voxel_num`` is read back from the device at most once per tensor object and version (the
  reference indexes a CUDA tensor five times per call, each a device->host sync);
* backward is one gather kernel (no boolean-mask indexing / nonzero sync, voxel_pooling.py:58-69).
"""
import weakref

import torch
from torch.autograd import Function

from ... import _lib, hip_ops

_MODE = "planned"
# plans of the operator-level calls without gradient, one per (device, stream, sizes): geom_xyz depends only on the
# camera calibration, so consecutive frames of a roadside camera hand in the same tensor content and the cached build
# (VoxelPlan(cached=True)) skips the rebuild on the device.  Keyed by stream as well, so that frames in flight on
# different streams never share a plan buffer.
_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 8
CACHE_PLANS = True


def set_mode(mode):
    """'planned' (deterministic, default) or 'atomic' (reference-faithful float atomics)."""
    global _MODE
    if mode not in ("planned", "atomic"):
        raise ValueError("mode must be 'planned' or 'atomic'")
    _MODE = mode


def get_mode():
    return _MODE


_VOXEL_NUM_CACHE = {}


def _voxel_num_ints(voxel_num):
    """(X, Y, Z) as python ints; a device tensor costs one sync per (tensor object, version)."""
    if isinstance(voxel_num, torch.Tensor):
        if voxel_num.numel() != 3:
            raise RuntimeError("voxel_num must have 3 elements")
        if voxel_num.is_cuda:
            # keyed by the tensor OBJECT (weak reference) and its version: an address alone can be a new tensor in freed memory
            ent = _VOXEL_NUM_CACHE.get(id(voxel_num))
            if ent is not None and ent[0]() is voxel_num and ent[1] == voxel_num._version:
                return ent[2]
            if len(_VOXEL_NUM_CACHE) > 64:
                _VOXEL_NUM_CACHE.clear()
            hit = tuple(int(v) for v in voxel_num.tolist())
            _VOXEL_NUM_CACHE[id(voxel_num)] = (weakref.ref(voxel_num), voxel_num._version, hit)
            return hit
        return tuple(int(v) for v in voxel_num.tolist())
    x, y, z = voxel_num
    return int(x), int(y), int(z)


def _check_cuda(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDAtensor ")          # voxel_pooling_forward.cpp:12-13
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")  # data_ptr<T>() :30-33


class VoxelPlan:
    """CSR plan (voxel -> ascending point ids) for one ``geom_xyz``; reusable while the camera
    calibration (hence ``geom_xyz``) is unchanged.

    ``cached=True``: the plan keeps a copy of the ``geom_xyz`` it was built for and ``rebuild(geom_xyz)`` only
    redoes the work when the new tensor differs -- decided on the device by one compare kernel, so the call never
    synchronises and can sit inside a captured hipGraph (the build kernels then return at once)."""

    def __init__(self, geom_xyz, voxel_num, pos_memo=None, sort_segments=True, cached=False):
        _check_cuda(geom_xyz, "geom_xyz", torch.int32)
        assert geom_xyz.is_contiguous()
        assert not (cached and pos_memo is not None), "the cached build writes no pos_memo (training builds a fresh plan)"
        self.X, self.Y, self.Z = _voxel_num_ints(voxel_num)
        self.B = int(geom_xyz.shape[0])
        self.N = int(geom_xyz.numel() // (3 * self.B))
        self.cached = bool(cached)
        self.sort_segments = bool(sort_segments)
        lib = _lib.load()
        nbytes = lib.sgv3d_voxel_plan_bytes(self.B, self.N, self.X, self.Y)
        if nbytes == 0:
            raise RuntimeError("voxel plan: bad sizes")
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=geom_xyz.device)
        self.nbytes = nbytes
        if cached:
            with torch.cuda.device(geom_xyz.device):
                _lib.check(lib.sgv3d_voxel_plan_init(self.B, self.N, self.X, self.Y, self.buf.data_ptr(), nbytes,
                                                     _lib.stream_handle(geom_xyz.device)), "sgv3d_voxel_plan_init")
        self.rebuild(geom_xyz, pos_memo)

    def rebuild(self, geom_xyz, pos_memo=None):
        """(Re)build for ``geom_xyz`` [B, N, 3] on the current stream; a cached plan skips the work on the device when
        the tensor is bytewise the one it holds."""
        assert geom_xyz.is_contiguous() and geom_xyz.numel() == self.B * self.N * 3 and geom_xyz.dtype == torch.int32
        lib = _lib.load()
        with torch.cuda.device(geom_xyz.device), hip_ops.prof("voxel_plan_build_cached" if self.cached else "voxel_plan_build"):
            st = _lib.stream_handle(geom_xyz.device)
            if self.cached:
                rc = lib.sgv3d_voxel_plan_build_cached(self.B, self.N, self.X, self.Y, self.Z, geom_xyz.data_ptr(),
                                                       self.buf.data_ptr(), self.nbytes, 1 if self.sort_segments else 0, st)
            else:
                rc = lib.sgv3d_voxel_plan_build(self.B, self.N, self.X, self.Y, self.Z, geom_xyz.data_ptr(),
                                                _lib.ptr(pos_memo), self.buf.data_ptr(), self.nbytes,
                                                1 if self.sort_segments else 0, st)
        _lib.check(rc, "sgv3d_voxel_plan_build")
        return self

    def builds(self):
        """Number of real (re)builds of a cached plan so far (reads the plan header: synchronises)."""
        assert self.cached
        off = _lib.load().sgv3d_voxel_plan_stats_offset(self.B, self.N, self.X, self.Y)
        return int(self.buf[off:off + 44].view(torch.int32)[10].item())

    def _workspace(self, C):
        nws = _lib.load().sgv3d_voxel_pooling_workspace_bytes(self.B, self.N, int(C))
        return torch.empty(nws, dtype=torch.uint8, device=self.buf.device), nws

    def pool(self, input_features, out=None, out_bf16_ld=0):
        """input_features f32 (or, bf16 compute mode, bf16) [B, N, C] -> f32 [B, Y, X, C] (fully written).
        ``out_bf16_ld`` (bf16 features only): the pooled map is written as bf16 [B, Y, X, out_bf16_ld] with zeroed padding
        channels -- the layout the bf16 BEV trunk reads."""
        bf16 = input_features.dtype == torch.bfloat16
        _check_cuda(input_features, "input_features", torch.bfloat16 if bf16 else torch.float32)
        assert input_features.is_contiguous() and (bf16 or not out_bf16_ld)
        C = int(input_features.shape[-1])
        assert input_features.numel() == self.B * self.N * C
        if out is None:
            out = torch.empty(self.B, self.Y, self.X, int(out_bf16_ld) or C, dtype=torch.bfloat16 if out_bf16_ld else torch.float32,
                              device=input_features.device)
        ws, nws = self._workspace(C)
        lib = _lib.load()
        with torch.cuda.device(input_features.device), hip_ops.prof("voxel_pooling_planned"):
            if bf16:
                rc = lib.sgv3d_voxel_pooling_forward_planned_bf16(
                    self.B, self.N, C, self.X, self.Y, self.buf.data_ptr(), input_features.data_ptr(), out.data_ptr(),
                    int(out_bf16_ld), ws.data_ptr(), nws, _lib.stream_handle(input_features.device))
            else:
                rc = lib.sgv3d_voxel_pooling_forward_planned(
                    self.B, self.N, C, self.X, self.Y, self.buf.data_ptr(), input_features.data_ptr(),
                    out.data_ptr(), ws.data_ptr(), nws, _lib.stream_handle(input_features.device))
        _lib.check(rc, "sgv3d_voxel_pooling_forward_planned")
        return out

    def lift_splat(self, prob, context, out=None, out_bf16_ld=0):
        """Fused path: prob f32 [B, D, P], context f32 [B, P, C] -> [B, Y, X, C] (``out_bf16_ld``: bf16 rows of that many
        channels, padding zeroed -- the bf16-activation hand-off of ``pool``)."""
        B, D, P = (int(s) for s in prob.shape)
        C = int(context.shape[-1])
        assert B == self.B and D * P == self.N and context.shape[:2] == (B, P)
        assert prob.is_contiguous() and context.is_contiguous() and prob.dtype == torch.float32
        ws, nws = self._workspace(C)
        if out_bf16_ld:
            assert context.dtype in (torch.float32, torch.bfloat16)
            if out is None:
                out = torch.empty(B, self.Y, self.X, int(out_bf16_ld), dtype=torch.bfloat16, device=context.device)
            with torch.cuda.device(context.device), hip_ops.prof("lift_splat_planned"):
                rc = _lib.load().sgv3d_lift_splat_planned_bf16out(B, D, P, C, self.X, self.Y, self.buf.data_ptr(), prob.data_ptr(),
                                                                 context.data_ptr(), 1 if context.dtype == torch.bfloat16 else 0,
                                                                 out.data_ptr(), int(out_bf16_ld), ws.data_ptr(),
                                                                 nws, _lib.stream_handle(context.device))
            _lib.check(rc, "sgv3d_lift_splat_planned_bf16out")
            return out
        assert context.dtype == torch.float32, "bf16 context rows come with the bf16 hand-off output (out_bf16_ld)"
        if out is None:
            out = context.new_empty(B, self.Y, self.X, C)
        with torch.cuda.device(context.device), hip_ops.prof("lift_splat_planned"):
            rc = _lib.load().sgv3d_lift_splat_planned(B, D, P, C, self.X, self.Y, self.buf.data_ptr(),
                                                     prob.data_ptr(), context.data_ptr(), out.data_ptr(),
                                                     ws.data_ptr(), nws, _lib.stream_handle(context.device))
        _lib.check(rc, "sgv3d_lift_splat_planned")
        return out


class VoxelPooling(Function):
    @staticmethod
    def forward(ctx, geom_xyz: torch.Tensor, input_features: torch.Tensor,
                voxel_num) -> torch.Tensor:
        """Forward function for `voxel pooling.

        Args:
            geom_xyz (Tensor): int32 voxel coord of each frustum point, shape [B, ..., 3].
            input_features (Tensor): float32 feature of each point, shape [B, ..., C].
            voxel_num (Tensor | sequence): number of voxels per dim (X, Y, Z).

        Returns:
            Tensor: (B, C, Y, X) bev feature map (a permuted view of an NHWC buffer).
        """
        assert geom_xyz.is_contiguous()                      # voxel_pooling.py:25
        assert input_features.is_contiguous()                # voxel_pooling.py:26
        _check_cuda(geom_xyz, "geom_xyz", torch.int32)
        _check_cuda(input_features, "input_features", torch.float32)
        ctx.mark_non_differentiable(geom_xyz)                # voxel_pooling.py:28
        features_shape = input_features.shape
        geom_xyz = geom_xyz.reshape(geom_xyz.shape[0], -1, geom_xyz.shape[-1])
        input_features = input_features.reshape(geom_xyz.shape[0], -1, input_features.shape[-1])
        assert geom_xyz.shape[1] == input_features.shape[1]  # voxel_pooling.py:33
        batch_size, num_points, num_channels = (int(s) for s in input_features.shape)
        X, Y, Z = _voxel_num_ints(voxel_num)
        needs_grad = ctx.needs_input_grad[1]
        pos_memo = None
        if needs_grad:
            pos_memo = torch.full((batch_size, num_points, 3), -1, dtype=torch.int32,
                                  device=geom_xyz.device)   # voxel_pooling.py:40
        lib = _lib.load()
        if _MODE == "atomic":
            # :37-38; filled by a kernel (new_full), not zero_(): inside a captured hipGraph zero_() becomes a memset
            # node, which faults on replay on this ROCm once the host allocates between replays
            output_features = input_features.new_full((batch_size, Y, X, num_channels), 0.0)
            with torch.cuda.device(input_features.device), hip_ops.prof("voxel_pooling_atomic"):
                rc = lib.sgv3d_voxel_pooling_forward_atomic(batch_size, num_points, num_channels, X, Y, Z,
                                                            geom_xyz.data_ptr(), input_features.data_ptr(),
                                                            output_features.data_ptr(), _lib.ptr(pos_memo),
                                                            _lib.stream_handle(input_features.device))
            _lib.check(rc, "sgv3d_voxel_pooling_forward_atomic")
        elif (CACHE_PLANS and not needs_grad and num_channels % 4 == 0 and 24 <= num_channels <= 256
              and input_features.data_ptr() % 16 == 0):
            # The library's own entry for the reference wrapper (level 1 of INTEGRATION.md) does everything an inference call
            # needs in 2-3 launches: it keeps a plan per (device, stream, sizes), compares geom_xyz with the tensor that plan was
            # built for ON THE DEVICE, sums with the deterministic gather while the geometry is unchanged (a roadside camera) and
            # with the reference's scatter the one call it is not.  The ``_fresh`` form writes every row of the map (empty
            # voxels as zeros), so the 21 MB zero fill of voxel_pooling.py:37-38 is not launched: torch.empty.
            # Calls that need a gradient (training: ida / bda augmentation moves geom_xyz every step) stay on the VoxelPlan
            # branch below -- always the deterministic gather, bit-reproducible run to run, which an alternation between
            # gather and scatter would not be.
            output_features = input_features.new_empty((batch_size, Y, X, num_channels))
            with torch.cuda.device(input_features.device), hip_ops.prof("voxel_pooling_level1"):
                rc = lib.sgv3d_voxel_pooling_forward_fresh(batch_size, num_points, num_channels, X, Y, Z, geom_xyz.data_ptr(),
                                                           input_features.data_ptr(), output_features.data_ptr(), None,
                                                           _lib.stream_handle(input_features.device))
            _lib.check(rc, "sgv3d_voxel_pooling_forward_fresh")
        elif needs_grad or not CACHE_PLANS or geom_xyz.data_ptr() % 16 != 0:
            # (the cached build compares geom_xyz with 16-byte loads: a contiguous slice such as geom[1:] whose storage
            # offset is not a multiple of 16 bytes -- accepted by the reference extension -- takes the uncached build)
            plan = VoxelPlan(geom_xyz, (X, Y, Z), pos_memo=pos_memo)
            output_features = plan.pool(input_features)
        else:
            dev = geom_xyz.device
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, batch_size, num_points, X, Y, Z)
            plan = _PLAN_CACHE.pop(key, None)
            if plan is None:
                while len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
                    _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
                plan = VoxelPlan(geom_xyz, (X, Y, Z), cached=True)
            else:
                plan.rebuild(geom_xyz)
            _PLAN_CACHE[key] = plan                             # most recently used last
            output_features = plan.pool(input_features)
        if needs_grad:
            ctx.save_for_backward(pos_memo)
            ctx.features_shape = features_shape
        return output_features.permute(0, 3, 1, 2)           # voxel_pooling.py:55

    @staticmethod
    def backward(ctx, grad_output_features):
        (pos_memo,) = ctx.saved_tensors
        shape = ctx.features_shape
        B, N = int(pos_memo.shape[0]), int(pos_memo.shape[1])
        C = int(shape[-1])
        g = grad_output_features
        if g.dtype != torch.float32:
            g = g.float()
        grad_input = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
        sb, sc, sy, sx = (int(s) for s in g.stride())
        with torch.cuda.device(g.device):
            rc = _lib.load().sgv3d_voxel_pooling_backward(B, N, C, pos_memo.data_ptr(), g.data_ptr(),
                                                         sb, sc, sy, sx, grad_input.data_ptr(),
                                                         _lib.stream_handle(g.device))
        _lib.check(rc, "sgv3d_voxel_pooling_backward")
        return None, grad_input.reshape(shape), None         # voxel_pooling.py:69


def _voxel_pooling_inference(geom_xyz, input_features, voxel_num):
    """The operator for a call that needs no gradient, with the host work cut to what the call needs: the same checks as
    ``VoxelPooling.forward`` (voxel_pooling.py:25-33), one output allocation, ONE library call
    (``sgv3d_voxel_pooling_forward_fresh``: plan per (device, stream, sizes) inside the library, every row of the map
    written).  At cfg-2 the gather is ~26 us of GPU time; the autograd ``Function.apply`` round trip, two context managers
    and two reshapes around it were 40 us of Python.  None: a case this path does not cover (the caller falls back)."""
    if not (geom_xyz.is_cuda and input_features.is_cuda and geom_xyz.dtype == torch.int32 and input_features.dtype == torch.float32):
        return None                                          # (VoxelPooling.forward raises the reference's errors)
    assert geom_xyz.is_contiguous()                          # voxel_pooling.py:25
    assert input_features.is_contiguous()                    # voxel_pooling.py:26
    num_channels = int(input_features.shape[-1])
    batch_size = int(geom_xyz.shape[0])
    num_points = geom_xyz.numel() // (3 * batch_size) if batch_size and int(geom_xyz.shape[-1]) == 3 else -1
    if (num_points <= 0 or input_features.numel() != batch_size * num_points * num_channels or num_channels % 4
            or not 24 <= num_channels <= 256 or input_features.data_ptr() % 16):
        return None
    X, Y, Z = _voxel_num_ints(voxel_num)
    dev = input_features.device
    output_features = torch.empty((batch_size, Y, X, num_channels), dtype=torch.float32, device=dev)
    if hip_ops.PROFILE is not None or dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev), hip_ops.prof("voxel_pooling_level1"):
            rc = _lib.load().sgv3d_voxel_pooling_forward_fresh(batch_size, num_points, num_channels, X, Y, Z, geom_xyz.data_ptr(),
                                                               input_features.data_ptr(), output_features.data_ptr(), None,
                                                               torch.cuda.current_stream(dev).cuda_stream)
    else:
        rc = _lib.load().sgv3d_voxel_pooling_forward_fresh(batch_size, num_points, num_channels, X, Y, Z, geom_xyz.data_ptr(),
                                                           input_features.data_ptr(), output_features.data_ptr(), None,
                                                           torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        _lib.check(rc, "sgv3d_voxel_pooling_forward_fresh")
    return output_features.permute(0, 3, 1, 2)               # voxel_pooling.py:55


def voxel_pooling(geom_xyz, input_features, voxel_num):
    """``voxel_pooling(geom_xyz, input_features, voxel_num) -> [B, C, Y, X]`` -- the reference's operator
    (ops/voxel_pooling/voxel_pooling.py:72, there ``VoxelPooling.apply``).  Calls that need a gradient go through
    ``VoxelPooling`` (autograd contract of voxel_pooling.py:10-69); the others through the lean inference path."""
    if (_MODE == "planned" and CACHE_PLANS and torch.is_tensor(input_features) and torch.is_tensor(geom_xyz)
            and not (input_features.requires_grad and torch.is_grad_enabled())):
        out = _voxel_pooling_inference(geom_xyz, input_features, voxel_num)
        if out is not None:
            return out
    return VoxelPooling.apply(geom_xyz, input_features, voxel_num)
