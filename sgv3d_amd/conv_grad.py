"""Backward of the NHWC convolutions (the training step of
exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:224-240 asks cuDNN for these through autograd).

* weight gradient: ``conv_wgrad_kernel`` (csrc/conv_wgrad.hip), fp32 MFMA, pixel-split with a fixed-order reduce;
* data gradient: a stride-1 convolution of the upstream gradient with the flipped, transposed weights through the
  forward kernels (implicit GEMM or Winograd, autotuned like every other layer); strided layers first spread the
  gradient over a zero-filled map (1x1 layers convolve at the coarse resolution and spread afterwards; kernel == stride
  layers are the transposed-convolution kernel with the same weights);
* ``conv2d`` is the autograd function tying the three together for NHWC tensors.

There is no CPU fallback: every function raises if the HIP library is missing.
"""
import ctypes

import torch

from . import _lib, grad_slots, pack_cache
from .hip_ops import ConvDesc, PackedConv, prof, _st

__all__ = ['conv2d_backward_weight', 'conv2d_backward_weight_bf16', 'conv2d_backward_weight_batched', 'conv2d_backward_data', 'zero_insert', 'conv2d', 'conv_transpose2d',
           'multi_conv2d', 'thin_conv3x3_forward_batched', 'thin_conv3x3_backward_batched', 'multi_thin_conv2d', 'sliced_thin_conv2d', 'cat_params', 'thin_conv_eligible']


def _out_hw(h, w, k, stride, pad, dil):
    return ((h + 2 * pad - dil * (k[0] - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k[1] - 1) - 1) // stride + 1)


def _dw_buffer(out, shape, device):
    if out is None:
        return torch.empty(*shape, dtype=torch.float32, device=device)
    assert tuple(out.shape) == tuple(shape) and out.is_contiguous() and out.dtype == torch.float32 and out.device == device
    return out


def conv2d_backward_weight(x, dy, kernel, stride=1, pad=0, dil=1, *, cin=None, cout=None, x_coff=0, y_coff=0, split=0, tile=0, out=None):
    """dW (OIHW [cout, cin, kh, kw]) of ``y = conv2d(x, W)`` for NHWC ``x`` [B, H, W, x_ld] and ``dy`` [B, OH, OW, y_ld];
    channel windows [x_coff, x_coff + cin) / [y_coff, y_coff + cout) default to the whole tensors.  ``out``: a contiguous
    f32 [cout, cin, kh, kw] tensor to write (a gradient slot of the flat buckets, grad_slots.claim)."""
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    B, H, W, x_ld = (int(v) for v in x.shape)
    _, OH, OW, y_ld = (int(v) for v in dy.shape)
    cin = x_ld - x_coff if cin is None else int(cin)
    cout = y_ld - y_coff if cout is None else int(cout)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype == torch.float32 and x.is_cuda
    assert (OH, OW) == _out_hw(H, W, (kh, kw), stride, pad, dil) and int(dy.shape[0]) == B
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = B, H, W, cin, OH, OW, cout
    d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, int(stride), int(pad), int(dil)
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, int(x_coff), y_ld, int(y_coff)
    d.tile = int(tile)
    lib = _lib.load()
    if (not tile and not split and kh == 3 and kw == 3 and stride == 1 and dil == 1 and cout <= 4 and y_ld <= 4):
        # 1..4 output channels (the final layers of the CenterHead branches): the vector-ALU kernel, not an MFMA tile of padding
        nws = lib.sgv3d_conv2d_backward_weight_thin_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
        dw = _dw_buffer(out, (cout, cin, kh, kw), x.device)
        with torch.cuda.device(x.device), prof("conv_wgrad_thin", 2.0 * B * OH * OW * cout * cin * kh * kw, 4.0 * (B * H * W * cin + B * OH * OW * cout + cout * cin * kh * kw)):
            rc = lib.sgv3d_conv2d_backward_weight_thin(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nws, _st(x))
        _lib.check(rc, "sgv3d_conv2d_backward_weight_thin")
        return dw
    from . import hip_ops
    if (hip_ops.MFMA_BF16 and hip_ops.TRAIN_BF16_WGRAD and not tile and not split and _bf16_wgrad_ok(x, dy, cin, cout, x_coff, y_coff)):
        # bf16 compute mode (mixed-precision training): products on the bf16 matrix cores, f32 tensors and accumulation
        return conv2d_backward_weight_bf16(x, dy, (kh, kw), stride, pad, dil, cin=cin, cout=cout, x_coff=x_coff, y_coff=y_coff, out=out)
    if not tile and not split:
        tile, split = _wgrad_choice(lib, d, x, dy)       # first-call measurement per layer shape (0, 0 = the library's rule)
        d.tile = int(tile)
    nws = lib.sgv3d_conv2d_backward_weight_workspace_bytes(ctypes.byref(d), int(split))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    dw = _dw_buffer(out, (cout, cin, kh, kw), x.device)
    name = "conv_wgrad"
    if hip_ops.PROFILE_DETAIL:
        name += f"|{B}x{H}x{W}x{cin}->{cout} k{kh} s{stride} d{dil} tile{int(tile)} split{int(split)}"
    with torch.cuda.device(x.device), prof(name, 2.0 * B * OH * OW * cout * cin * kh * kw, 4.0 * (B * H * W * cin + B * OH * OW * cout + cout * cin * kh * kw)):
        rc = lib.sgv3d_conv2d_backward_weight(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(split),
                                              ws.data_ptr(), nws, _st(x))
    _lib.check(rc, "sgv3d_conv2d_backward_weight")
    return dw


def _bf16_wgrad_ok(x, dy, cin, cout, x_coff, y_coff):
    x_ld, y_ld = int(x.shape[-1]), int(dy.shape[-1])
    return (cin % 4 == 0 and cout % 4 == 0 and x_ld % 4 == 0 and y_ld % 4 == 0 and x_coff % 4 == 0 and y_coff % 4 == 0
            and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and min(cin, cout) >= 16)


def conv2d_backward_weight_bf16(x, dy, kernel, stride=1, pad=0, dil=1, *, cin=None, cout=None, x_coff=0, y_coff=0, split=0, tile=0, out=None):
    """``conv2d_backward_weight`` with the products on the bf16 matrix cores (sgv3d_conv2d_backward_weight_bf16): f32 tensors in
    and out, operands rounded to bf16 while staging, f32 accumulation, fixed-order pixel-split reduce.  ``tile``: 0 = measured per
    layer shape (first call) / the library's rule, 1 = 64 x 64, 4 = 128 x 128 (one tap per workgroup, ``split`` = pixel ranges), 6 = the
    all-taps kernel (3x3 / stride 1: a workgroup owns a 64 x 64 tile for all nine taps and walks down a column of the map;
    ``split`` = row chunks per column, 0 = rule)."""
    from . import hip_ops
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    B, H, W, x_ld = (int(v) for v in x.shape)
    _, OH, OW, y_ld = (int(v) for v in dy.shape)
    cin = x_ld - x_coff if cin is None else int(cin)
    cout = y_ld - y_coff if cout is None else int(cout)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype == torch.float32 and x.is_cuda
    assert (OH, OW) == _out_hw(H, W, (kh, kw), stride, pad, dil) and int(dy.shape[0]) == B
    if not _bf16_wgrad_ok(x, dy, cin, cout, x_coff, y_coff):
        raise _lib.SGV3DError("the bf16 weight-gradient kernel needs channel counts, strides and offsets that are multiples of 4 "
                              "(at least 16 channels on both sides) and 16-byte aligned tensors")
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = B, H, W, cin, OH, OW, cout
    d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, int(stride), int(pad), int(dil)
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, int(x_coff), y_ld, int(y_coff)
    d.tile = int(tile)
    lib = _lib.load()
    if not tile and not split:
        tile, split = _wgrad_choice(lib, d, x, dy, bf16=True)
        d.tile = int(tile)
    if int(tile) == 6:              # all nine taps per workgroup (csrc/conv_wgrad3x3_bf16.hip); split = row chunks per column
        d.tile = 0
        nws = lib.sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes(ctypes.byref(d), 1, int(split))
        if nws == 0:
            raise _lib.SGV3DError("the all-taps bf16 weight-gradient kernel takes 3x3 / stride-1 layers (dilation <= 20) only")
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        dw = _dw_buffer(out, (cout, cin, kh, kw), x.device)
        name = "conv_wgrad_bf16_alltaps"
        if hip_ops.PROFILE_DETAIL:
            name += f"|{B}x{H}x{W}x{cin}->{cout} k{kh} s{stride} d{dil} tile6 split{int(split)}"
        with torch.cuda.device(x.device), prof(name, 2.0 * B * OH * OW * cout * cin * kh * kw, 4.0 * (B * H * W * cin + B * OH * OW * cout + cout * cin * kh * kw)):
            rc = lib.sgv3d_conv2d_backward_weight_bf16_alltaps(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(split),
                                                               ws.data_ptr(), nws, _st(x))
        _lib.check(rc, "sgv3d_conv2d_backward_weight_bf16_alltaps")
        return dw
    nws = lib.sgv3d_conv2d_backward_weight_bf16_workspace_bytes(ctypes.byref(d), int(split))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    dw = _dw_buffer(out, (cout, cin, kh, kw), x.device)
    name = "conv_wgrad_bf16"
    if hip_ops.PROFILE_DETAIL:
        name += f"|{B}x{H}x{W}x{cin}->{cout} k{kh} s{stride} d{dil} tile{int(tile)} split{int(split)}"
    with torch.cuda.device(x.device), prof(name, 2.0 * B * OH * OW * cout * cin * kh * kw, 4.0 * (B * H * W * cin + B * OH * OW * cout + cout * cin * kh * kw)):
        rc = lib.sgv3d_conv2d_backward_weight_bf16(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(split),
                                                   ws.data_ptr(), nws, _st(x))
    _lib.check(rc, "sgv3d_conv2d_backward_weight_bf16")
    return dw


def conv2d_backward_weight_batched(x, dys, pad=1, *, cin=None, cout=None, split=0, outs=None):
    """[dW_i] of n 3x3 / stride-1 convolutions that read the same NHWC ``x``: one launch of the all-taps kernel
    (sgv3d_conv2d_backward_weight_batched).  ``dys``: n contiguous NHWC tensors of one shape."""
    B, H, W, x_ld = (int(v) for v in x.shape)
    _, OH, OW, y_ld = (int(v) for v in dys[0].shape)
    cin = x_ld if cin is None else int(cin)
    cout = y_ld if cout is None else int(cout)
    n = len(dys)
    assert x.is_contiguous() and x.dtype == torch.float32 and all(d.is_contiguous() and d.dtype == torch.float32 and d.shape == dys[0].shape for d in dys)
    assert (OH, OW) == _out_hw(H, W, (3, 3), 1, pad, 1) and 0 < n <= 48
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = B, H, W, cin, OH, OW, cout
    d.kh, d.kw, d.stride, d.pad, d.dil = 3, 3, 1, int(pad), 1
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, 0, y_ld, 0
    from . import hip_ops
    lib = _lib.load()
    if (hip_ops.MFMA_BF16 and hip_ops.TRAIN_BF16_WGRAD and not split and _bf16_wgrad_ok(x, dys[0], cin, cout, 0, 0)
            and all(t.data_ptr() % 16 == 0 for t in dys)):
        # mixed-precision step: the same n gradients on the bf16 matrix cores, one launch (blockIdx.z = problem)
        d.tile = 0
        if hip_ops.WGRAD_BF16_ALLTAPS:
            nws = lib.sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes(ctypes.byref(d), n, 0)
            ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
            dws = [_dw_buffer(None if outs is None else outs[i], (cout, cin, 3, 3), x.device) for i in range(n)]
            dyp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dys])
            dwp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dws])
            with torch.cuda.device(x.device), prof("conv_wgrad_bf16_alltaps", 2.0 * n * B * OH * OW * cout * cin * 9, 4.0 * (B * H * W * cin + n * (B * OH * OW * cout + cout * cin * 9))):
                rc = lib.sgv3d_conv2d_backward_weight_bf16_alltaps_batched(ctypes.byref(d), x.data_ptr(), dyp, dwp, n, 0, ws.data_ptr(), nws, _st(x))
            _lib.check(rc, "sgv3d_conv2d_backward_weight_bf16_alltaps_batched")
            return dws
        tiles = -(-cout // 64) * -(-cin // 64) * 9 * n
        sp = max(1, min(-(-1024 // tiles), (B * OH * OW) // 256))
        nws = lib.sgv3d_conv2d_backward_weight_bf16_batched_workspace_bytes(ctypes.byref(d), n, int(sp))
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
        dws = [_dw_buffer(None if outs is None else outs[i], (cout, cin, 3, 3), x.device) for i in range(n)]
        dyp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dys])
        dwp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dws])
        with torch.cuda.device(x.device), prof("conv_wgrad_bf16", 2.0 * n * B * OH * OW * cout * cin * 9, 4.0 * (B * H * W * cin + n * (B * OH * OW * cout + cout * cin * 9))):
            rc = lib.sgv3d_conv2d_backward_weight_bf16_batched(ctypes.byref(d), x.data_ptr(), dyp, dwp, n, int(sp), ws.data_ptr(), nws, _st(x))
        _lib.check(rc, "sgv3d_conv2d_backward_weight_bf16_batched")
        return dws
    d.tile = 5
    if not split:                       # ~3 workgroups per CU over all problems, at least four row segments per workgroup
        units = B * OH * -(-OW // 32)
        tiles = -(-cout // 64) * -(-cin // 64) * n
        split = max(1, min(-(-768 // tiles), units // 4))
    nws = lib.sgv3d_conv2d_backward_weight_batched_workspace_bytes(ctypes.byref(d), n, int(split))
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
    dws = [_dw_buffer(None if outs is None else outs[i], (cout, cin, 3, 3), x.device) for i in range(n)]
    dyp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dys])
    dwp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dws])
    with torch.cuda.device(x.device), prof("conv_wgrad", 2.0 * n * B * OH * OW * cout * cin * 9, 4.0 * (B * H * W * cin + n * (B * OH * OW * cout + cout * cin * 9))):
        rc = lib.sgv3d_conv2d_backward_weight_batched(ctypes.byref(d), x.data_ptr(), dyp, dwp, n, int(split), ws.data_ptr(), nws, _st(x))
    _lib.check(rc, "sgv3d_conv2d_backward_weight_batched")
    return dws


_WGRAD_DB = {}


def _wgrad_choice(lib, d, x, dy, bf16=False):
    """(tile, split) of the weight-gradient kernel for this layer shape: the library's rule and a handful of alternatives
    (the four tile shapes; half / twice / four times the rule's pixel split) timed once on the real tensors.  Every choice
    sums each tile's pixel ranges in a fixed order, so results differ only by the association of that sum.  ``bf16``: the
    bf16-MFMA kernel (tiles 1 = 64 x 64 and 4 = 128 x 128 only)."""
    from . import hip_ops
    key = (d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout, d.kh, d.kw, d.stride, d.pad, d.dil, d.x_ld, d.y_ld) + (("bf16",) if bf16 else ())
    if key in _WGRAD_DB:
        return _WGRAD_DB[key]
    sig = "wgrad|" + "x".join(str(v) for v in key)      # committed choices (tune/gfx950_*train*.json): no first-call timing, same kernels every run
    ws_bytes = lib.sgv3d_conv2d_backward_weight_bf16_workspace_bytes if bf16 else lib.sgv3d_conv2d_backward_weight_workspace_bytes
    launch = lib.sgv3d_conv2d_backward_weight_bf16 if bf16 else lib.sgv3d_conv2d_backward_weight
    if not hip_ops.AUTOTUNE:
        return 0, 0                                          # the library's own rule, whatever a tune DB holds
    if sig in hip_ops.TUNE_DB and not (sig in hip_ops._COMMITTED_SIGS and not hip_ops._is_gfx950(x.device)):
        _WGRAD_DB[key] = tuple(hip_ops.TUNE_DB[sig])
        return _WGRAD_DB[key]
    if torch.cuda.is_current_stream_capturing():
        return 0, 0
    pixels = d.batch * d.out_h * d.out_w
    cands = [(0, 0)]
    for t in ((1, 4) if bf16 else (1, 2, 3, 4)):
        tiles = -(-d.cout // (128 if t > 2 else 64)) * -(-d.cin // (128 if t in (2, 4) else 64)) * d.kh * d.kw
        base = max(1, min(max(256, min(1536, tiles * 64)) // max(tiles, 1), pixels // 256))
        for f in (0.5, 1, 2, 4):
            sp = int(max(1, min(base * f, pixels // 128)))
            if (t, sp) not in cands:
                cands.append((t, sp))
    if (not bf16 and d.kh == 3 and d.kw == 3 and d.stride == 1 and d.dil == 1 and d.cin % 4 == 0 and d.cout % 4 == 0 and d.x_coff % 4 == 0
            and d.y_coff % 4 == 0 and d.x_ld % 4 == 0 and d.y_ld % 4 == 0):
        # the all-taps kernel (tile 5): work units = (image, output row, 32-pixel segment); the rule's split and neighbours
        units = d.batch * d.out_h * -(-d.out_w // 32)
        t2 = -(-d.cout // 64) * -(-d.cin // 64)
        base = max(1, min(-(-512 // t2), units // 4))
        for f in (0.5, 1, 2, 4):
            sp = int(max(1, min(base * f, units)))
            if (5, sp) not in cands:
                cands.append((5, sp))
    if (bf16 and hip_ops.WGRAD_BF16_ALLTAPS and d.kh == 3 and d.kw == 3 and d.stride == 1 and d.dil <= 20):
        # the all-taps bf16 kernel (tile 6): split = row chunks per column of the map; the rule (0) and fixed counts
        for sp in (0, 1, 2, 4, 8):
            cands.append((6, sp))
    dw = torch.empty(d.cout, d.cin, d.kh, d.kw, dtype=torch.float32, device=x.device)
    best, best_t = (0, 0), None
    with torch.cuda.device(x.device):
        for t, sp in cands:
            d.tile = t
            if t == 6:
                d.tile = 0
                nws = lib.sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes(ctypes.byref(d), 1, sp)
                if nws == 0 or nws > (2 << 30):
                    continue
                ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
                run = lambda: lib.sgv3d_conv2d_backward_weight_bf16_alltaps(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), sp,
                                                                            ws.data_ptr(), nws, _st(x))
            else:
                nws = ws_bytes(ctypes.byref(d), sp)
                if nws > (2 << 30):
                    continue
                ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=x.device)
                run = lambda: launch(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), sp, ws.data_ptr(), nws, _st(x))
            if run() != 0:
                continue
            nrep = max(3, hip_ops.TUNE_REPEATS + 1)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nrep + 1)]
            evs[0].record()
            for r in range(nrep):
                run()
                evs[r + 1].record()
            evs[-1].synchronize()
            dt = min(evs[r].elapsed_time(evs[r + 1]) for r in range(nrep))
            if best_t is None or dt < best_t:
                best, best_t = (t, sp), dt
    d.tile = 0
    _WGRAD_DB[key] = best
    hip_ops.TUNE_DB[sig] = list(best)
    hip_ops._COMMITTED_SIGS.discard(sig)
    return best


def zero_insert(x, stride, out_hw):
    """NHWC map with ``x`` on every ``stride``-th pixel and zeros elsewhere."""
    B, H, W, C = (int(v) for v in x.shape)
    assert x.is_contiguous() and x.dtype == torch.float32 and C % 4 == 0
    y = torch.empty(B, out_hw[0], out_hw[1], C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), prof("zero_insert"):
        rc = _lib.load().sgv3d_zero_insert(B, H, W, C, int(stride), int(out_hw[0]), int(out_hw[1]), x.data_ptr(),
                                           y.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_zero_insert")
    return y


def _packed_call(weight, key, make, x, **kw):
    """``make(weight)(x, **kw)`` -- through the layer's kept PackedConv while the optimiser's pack-cache window is open
    (sgv3d_amd/pack_cache.py: packed forms refreshed by one launch per step), packed from the current weights otherwise."""
    e = pack_cache.entry_for(weight, key, make)
    return e.call(x, **kw) if e is not None else make(weight.detach())(x, **kw)


def _rot180_transpose(w):
    """``w.flip(2, 3).transpose(0, 1).contiguous()`` of an OIHW f32 weight tensor in one launch."""
    w = w.contiguous()
    cout, cin, kh, kw = (int(v) for v in w.shape)
    out = torch.empty(cin, cout, kh, kw, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().sgv3d_weight_rot180_transpose(w.data_ptr(), cout, cin, kh, kw, out.data_ptr(), _st(w))
    _lib.check(rc, "sgv3d_weight_rot180_transpose")
    return out


def conv2d_backward_data(dy, weight, in_hw, stride=1, pad=0, dil=1, add_to=None):
    """dX (NHWC [B, H, W, cin]) of ``y = conv2d(x, weight)``; ``weight`` OIHW, ``dy`` NHWC [B, OH, OW, cout(+padding to 4)].
    ``add_to`` (stride 1 only): a tensor of dX's shape added in the convolution's epilogue (the sum of the data gradients of
    several consumers of one map without separate add launches)."""
    cout, cin, kh, kw = (int(v) for v in weight.shape)
    H, W = in_hw
    assert kh == kw, "square kernels only (every layer of the model)"
    if int(dy.shape[-1]) % 4:
        dy = torch.nn.functional.pad(dy, (0, 4 - int(dy.shape[-1]) % 4))
    if kh == stride and stride > 1 and pad == 0 and dil == 1:
        # kernel == stride ("patchify" convolutions of the necks): every input pixel belongs to exactly one output pixel,
        # so the data gradient is the transposed convolution with the same weight tensor read as [in = cout, out = cin, k, k]
        cp = int(dy.shape[-1])
        dx = _packed_call(weight, ('dgrad_patch', stride, cp), lambda w: PackedConv(w, stride=stride, transposed=True, cin_pad=cp), dy)
        ph, pw = H - int(dx.shape[1]), W - int(dx.shape[2])
        return dx if ph == 0 and pw == 0 else torch.nn.functional.pad(dx, (0, 0, 0, pw, 0, ph))   # rows the conv never read
    cp, rpad = int(dy.shape[-1]), dil * (kh - 1) - pad
    # the stride-1 convolution with the weights rotated by 180 degrees and in / out swapped ([cin, cout, kh, kw])
    make = lambda w: PackedConv(_rot180_transpose(w), stride=1, pad=rpad, dil=dil, cin_pad=cp, pad_out=True)
    conv = lambda x, **kw: _packed_call(weight, ('dgrad', rpad, dil, cp), make, x, **kw)
    if stride == 1:
        return conv(dy, residual=add_to)
    assert add_to is None, "add_to: stride-1 layers only"
    if kh == 1 and pad == 0:
        return zero_insert(conv(dy), stride, (H, W))                      # 1x1: convolve at the coarse resolution
    if stride == 2 and dil == 1:
        return _dgrad_stride2(dy, weight, (H, W), pad)
    hz, wz = H + 2 * pad - dil * (kh - 1), W + 2 * pad - dil * (kw - 1)
    return conv(zero_insert(dy, stride, (hz, wz)))


def _phase_1d(k, pad, par, n_in, n_out):
    """Taps and geometry of output parity ``par`` of a stride-2 data gradient along one axis:
    dX[2 i + par] = sum_t dY[i + q - t] w[r + 2 t].  Returns (tap indices in correlation order, low padding, outputs)."""
    r = (par + pad) % 2
    taps = list(range(r, k, 2))
    q = (par + pad - r) // 2
    T = len(taps)
    return taps[::-1], (T - 1) - q, (n_out - par + 1) // 2      # flipped taps: a cross-correlation reading dY[i - P + t']


def _dgrad_stride2(dy, weight, in_hw, pad):
    """Stride-2 data gradient as four stride-1 convolutions of ``dy`` (one per output parity, each with the taps of that
    parity: together exactly the multiplies of the direct form, 4x fewer than convolving a zero-filled map) whose
    results are interleaved by ``sgv3d_interleave_phases2``."""
    cout, cin, kh, kw = (int(v) for v in weight.shape)
    H, W = in_hw
    B, hc, wc, _ = (int(v) for v in dy.shape)
    outs, geo = [], []
    for py in range(2):
        ty, p_y, n_y = _phase_1d(kh, pad, py, hc, H)
        for px in range(2):
            tx, p_x, n_x = _phase_1d(kw, pad, px, wc, W)
            if not ty or not tx or n_y == 0 or n_x == 0:          # no tap has this parity: the phase is zero
                outs.append(torch.zeros(B, max(n_y, 1), max(n_x, 1), (cin + 3) // 4 * 4, dtype=torch.float32, device=dy.device))
                geo.append((max(n_y, 1), max(n_x, 1), 0, 0))
                continue
            # (taps are min, min + 2, ... in descending order: a strided slice and a flip -- indexing with the Python lists would build
            # index tensors on the host and copy them over, which a stream capture cannot record)
            # one symmetric padding that covers the low side of both axes and yields enough outputs on the high side
            need = lambda P, T, n_in, n_out: max(P, 0, n_out - n_in + T - 1 - P)
            pp = max(need(p_y, len(ty), hc, n_y), need(p_x, len(tx), wc, n_x))
            cp, ay, ax = int(dy.shape[-1]), min(ty), min(tx)
            # sub-kernel [cin, cout, Ty, Tx] of the phase
            make = lambda v, ay=ay, ax=ax, pp=pp: PackedConv(v[:, :, ay::2, ax::2].flip(2, 3).transpose(0, 1).contiguous(), stride=1, pad=pp,
                                                             cin_pad=cp, pad_out=True)
            o = _packed_call(weight, ('dgrad_s2', py, px, pad, pp, cp), make, dy)
            outs.append(o)
            geo.append((int(o.shape[1]), int(o.shape[2]), pp - p_y, pp - p_x))
    C = int(outs[0].shape[-1])
    dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
    ptrs = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs])
    arr = lambda i: (ctypes.c_int32 * 4)(*[g[i] for g in geo])
    with torch.cuda.device(dy.device), prof("interleave_phases"):
        rc = _lib.load().sgv3d_interleave_phases2(B, H, W, C, ptrs, arr(0), arr(1), arr(2), arr(3), dx.data_ptr(), _st(dy))
    _lib.check(rc, "sgv3d_interleave_phases2")
    return dx


def _pad4(x):
    c = int(x.shape[-1])
    return x if c % 4 == 0 else torch.nn.functional.pad(x, (0, 4 - c % 4))


class _Conv2dNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        cp = int(x.shape[-1])
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, dil, bias is not None)
        return _packed_call(weight, ('fwd', stride, pad, dil, cp, None if bias is None else bias.data_ptr()),
                            lambda w: PackedConv(w, stride=stride, pad=pad, dil=dil, shift=bias, cin_pad=cp), x)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.geom
        dy = dy.contiguous()
        cout, cin, kh, kw = (int(v) for v in weight.shape)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_backward_data(dy, weight, (int(x.shape[1]), int(x.shape[2])), stride, pad, dil)
            if int(dx.shape[-1]) != int(x.shape[-1]):
                dx = torch.nn.functional.pad(dx, (0, int(x.shape[-1]) - int(dx.shape[-1])))
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(x, dy, (kh, kw), stride, pad, dil, cin=cin, cout=cout, out=grad_slots.claim(weight))
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 1, 2))
        return dx, dw, db, None, None, None


class _MultiConv2dNHWC(torch.autograd.Function):
    """n 3x3 / stride-1 / pad-1 convolutions without bias of ONE input (the first layers of the CenterHead branches): forward and data
    gradients are the per-layer kernels, the n weight gradients are one batched launch."""
    @staticmethod
    def forward(ctx, x, *weights):
        ctx.save_for_backward(x, *weights)
        return tuple(PackedConv(w, stride=1, pad=1, cin_pad=int(x.shape[-1]))(x) for w in weights)

    @staticmethod
    def backward(ctx, *dys):
        x, *weights = ctx.saved_tensors
        cout, cin = int(weights[0].shape[0]), int(weights[0].shape[1])
        dys = [torch.zeros(x.shape[:3] + (cout,), dtype=x.dtype, device=x.device) if d is None else d.contiguous() for d in dys]
        dx = None
        if ctx.needs_input_grad[0]:
            for d, w in zip(dys, weights):                # (the running sum rides in each convolution's residual epilogue)
                dx = conv2d_backward_data(d, w, (int(x.shape[1]), int(x.shape[2])), 1, 1, 1, add_to=dx)
            if int(dx.shape[-1]) != int(x.shape[-1]):
                dx = torch.nn.functional.pad(dx, (0, int(x.shape[-1]) - int(dx.shape[-1])))
        dws = [None] * len(weights)
        if any(ctx.needs_input_grad[1:]):
            dws = conv2d_backward_weight_batched(x, dys, 1, cin=cin, cout=cout, outs=[grad_slots.claim(w) for w in weights])
        return (dx, *dws)


def thin_conv3x3_backward_batched(xs, dys, weights, pad=1, *, need_dx=True, need_dw=True, need_db=False, dw_outs=None, db_outs=None, dx_like=None):
    """The whole backward of n 3x3 / stride-1 convolutions with 1..4 output channels (sgv3d_conv3x3_thin_backward_batched): one launch
    per gradient kind and output-channel count instead of three launches per layer.  ``xs[i]`` NHWC f32 [B, H, W, cin] (one shape for
    all), ``dys[i]`` NHWC f32 [B, OH, OW, cout_i], ``weights[i]`` OIHW [cout_i, cin, 3, 3].  Returns (dxs | None, dws | None, dbs | None);
    ``dw_outs`` / ``db_outs``: per-layer output buffers or None entries (gradient slots of the flat buckets).  The ``xs`` may be the n
    equal channel slices of one contiguous map (``x_ld`` = its channel count); ``dx_like`` is then an empty map of that shape whose slices
    receive the data gradients."""
    n = len(xs)
    assert n == len(dys) == len(weights) and 0 < n <= 48
    B, H, W, cin = (int(v) for v in xs[0].shape)
    _, OH, OW, _ = (int(v) for v in dys[0].shape)
    dev = xs[0].device
    assert (OH, OW) == _out_hw(H, W, (3, 3), 1, pad, 1)
    couts = [int(w.shape[0]) for w in weights]
    dys = [d.contiguous() for d in dys]
    x_ld = _slice_ld(xs)
    for x, d, w, c in zip(xs, dys, weights, couts):
        assert tuple(d.shape) == (B, OH, OW, c) and d.dtype == torch.float32
        assert tuple(w.shape) == (c, cin, 3, 3) and w.is_contiguous() and w.dtype == torch.float32
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = B, H, W, cin, OH, OW, max(couts)
    d.kh, d.kw, d.stride, d.pad, d.dil = 3, 3, 1, int(pad), 1
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, 0, max(couts), 0
    lib = _lib.load()
    cc = (ctypes.c_int32 * n)(*couts)
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    dxs = None
    if need_dx:
        if dx_like is not None:                    # channel slices of ONE map shaped like the one the inputs are slices of
            assert x_ld > cin and tuple(dx_like.shape) == (B, H, W, x_ld) and dx_like.is_contiguous() and x_ld == n * cin
            dxs = [dx_like[..., i * cin:(i + 1) * cin] for i in range(n)]
        else:
            assert x_ld == cin, "channel-slice inputs: pass dx_like (the map to write the data gradients' slices into)"
            dxs = [torch.empty_like(x) for x in xs]
    dws = dbs = None
    ws, nws = None, 0
    if need_dw or need_db:
        nws = lib.sgv3d_conv3x3_thin_backward_batched_workspace_bytes(ctypes.byref(d), n, cc)
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
        if need_dw:
            dws = [_dw_buffer(None if dw_outs is None else dw_outs[i], (couts[i], cin, 3, 3), dev) for i in range(n)]
        if need_db:
            dbs = [_dw_buffer(None if db_outs is None else db_outs[i], (couts[i],), dev) for i in range(n)]
    px = B * OH * OW
    flops = sum(2.0 * px * c * cin * 9 for c in couts) * (int(bool(need_dx)) + int(bool(need_dw)))
    with torch.cuda.device(dev), prof("conv_thin_backward_batched", flops, 4.0 * n * (B * H * W * cin * (int(bool(need_dx)) + int(bool(need_dw or need_db))) + px * max(couts))):
        rc = lib.sgv3d_conv3x3_thin_backward_batched(
            ctypes.byref(d), n, cc, arr(xs), arr(dys), arr(weights), arr(dxs) if need_dx else None, arr(dws) if need_dw else None,
            arr(dbs) if need_db else None, _lib.ptr(ws), nws, _st(xs[0]))
    _lib.check(rc, "sgv3d_conv3x3_thin_backward_batched")
    return dxs, dws, dbs


def _slice_ld(xs):
    """Pixel stride (floats) of the NHWC f32 tensors ``xs``: contiguous tensors (stride = channel count) or equal channel slices of a
    contiguous map; all of one shape and one stride, 16-byte aligned."""
    B, H, W, cin = (int(v) for v in xs[0].shape)
    ld = int(xs[0].stride(2))
    for x in xs:
        assert x.shape == xs[0].shape and x.dtype == torch.float32 and x.is_cuda and x.data_ptr() % 16 == 0
        assert x.stride(3) == 1 and int(x.stride(2)) == ld and int(x.stride(1)) == W * ld and int(x.stride(0)) == H * W * ld, \
            "NHWC tensors or channel slices of one NHWC map"
    assert ld >= cin and ld % 4 == 0
    return ld


def thin_conv3x3_forward_batched(xs, weights, biases=None, pad=1):
    """[conv3x3(x, w, stride 1) + b] of n layers with 1..4 output channels and at most 64 input channels in one launch per
    output-channel count (sgv3d_conv3x3_thin_forward_batched, f32 arithmetic): NHWC f32 ``xs[i]`` [B, H, W, cin] -> [B, OH, OW, cout_i]."""
    n = len(xs)
    assert n == len(weights) and 0 < n <= 48
    B, H, W, cin = (int(v) for v in xs[0].shape)
    OH, OW = _out_hw(H, W, (3, 3), 1, pad, 1)
    dev = xs[0].device
    couts = [int(w.shape[0]) for w in weights]
    x_ld = _slice_ld(xs)
    for w, c in zip(weights, couts):
        assert tuple(w.shape) == (c, cin, 3, 3) and w.is_contiguous() and w.dtype == torch.float32
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = B, H, W, cin, OH, OW, max(couts)
    d.kh, d.kw, d.stride, d.pad, d.dil = 3, 3, 1, int(pad), 1
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, 0, max(couts), 0
    ys = [torch.empty(B, OH, OW, c, dtype=torch.float32, device=dev) for c in couts]
    cc = (ctypes.c_int32 * n)(*couts)
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    bl = None if biases is None else [None if b is None else b.detach().float().contiguous() for b in biases]
    with torch.cuda.device(dev), prof("conv_thin_forward_batched", sum(2.0 * B * OH * OW * c * cin * 9 for c in couts),
                                      4.0 * n * (B * H * W * cin + B * OH * OW * max(couts))):
        rc = _lib.load().sgv3d_conv3x3_thin_forward_batched(ctypes.byref(d), n, cc, arr(xs), arr(weights), None if bl is None else arr(bl), arr(ys),
                                                            _st(xs[0]))
    _lib.check(rc, "sgv3d_conv3x3_thin_forward_batched")
    return ys


def thin_conv_eligible(convs, xs):
    """Whether ``multi_thin_conv2d`` takes these nn.Conv2d layers on these inputs."""
    if not (1 < len(convs) <= 48 and len(convs) == len(xs)):
        return False
    cin = int(convs[0].weight.shape[1])
    return all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1
               and 1 <= int(c.weight.shape[0]) <= 4 and int(c.weight.shape[1]) == cin for c in convs) and cin % 4 == 0 and \
        all(x.shape == xs[0].shape and int(x.shape[-1]) == cin and x.dtype == torch.float32 for x in xs)


class _MultiThinConv2dNHWC(torch.autograd.Function):
    """n independent 3x3 / stride-1 / pad-1 convolutions with 1..4 output channels each (the final layers of the CenterHead
    branches): forward through the per-layer kernels, the 3 n gradient launches of the backward as one batched call."""
    @staticmethod
    def forward(ctx, n, *args):
        xs, weights, biases = args[:n], args[n:2 * n], args[2 * n:3 * n]
        ctx.n = n
        ctx.save_for_backward(*xs, *weights)
        ctx.has_bias = [b is not None for b in biases]
        ctx.bias_refs = biases                      # (parameters: only their gradient slots are looked up in backward)
        if int(xs[0].shape[-1]) <= 64 and all(x.dtype == torch.float32 and x.is_contiguous() for x in xs):
            # one launch per output-channel count instead of one MFMA launch per layer padded to 64 output columns
            return tuple(thin_conv3x3_forward_batched(list(xs), [w.detach() for w in weights], list(biases), 1))
        return tuple(PackedConv(w, stride=1, pad=1, shift=b, cin_pad=int(x.shape[-1]))(x) for x, w, b in zip(xs, weights, biases))

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        xs, weights = saved[:n], saved[n:2 * n]
        B, H, W, _ = (int(v) for v in xs[0].shape)
        dys = [torch.zeros(B, H, W, int(w.shape[0]), dtype=torch.float32, device=xs[0].device) if d is None else d for d, w in zip(dys, weights)]
        need_dx = any(ctx.needs_input_grad[1:1 + n])
        need_dw = any(ctx.needs_input_grad[1 + n:1 + 2 * n])
        need_db = any(h and g for h, g in zip(ctx.has_bias, ctx.needs_input_grad[1 + 2 * n:1 + 3 * n]))
        dxs, dws, dbs = thin_conv3x3_backward_batched(
            xs, dys, [w.detach() for w in weights], 1, need_dx=need_dx, need_dw=need_dw, need_db=need_db,
            dw_outs=[grad_slots.claim(w) for w in weights] if need_dw else None,
            db_outs=[grad_slots.claim(b) if h else None for b, h in zip(ctx.bias_refs, ctx.has_bias)] if need_db else None)
        none = [None] * n
        dbs = none if dbs is None else [g if h else None for g, h in zip(dbs, ctx.has_bias)]
        return (None, *(dxs or none), *(dws or none), *dbs)


class _SlicedThinConv2dNHWC(torch.autograd.Function):
    """The n thin layers of _MultiThinConv2dNHWC reading the n equal channel slices of ONE map [B, H, W, n * cin] (the hidden maps of
    the CenterHead branches kept as one tensor): the data gradients are written into the slices of one map of the same shape."""
    @staticmethod
    def forward(ctx, big, n, *args):
        weights, biases = args[:n], args[n:2 * n]
        cin = int(big.shape[-1]) // n
        ctx.n, ctx.cin = n, cin
        ctx.save_for_backward(big, *weights)
        ctx.has_bias = [b is not None for b in biases]
        ctx.bias_refs = biases
        xs = [big[..., i * cin:(i + 1) * cin] for i in range(n)]
        return tuple(thin_conv3x3_forward_batched(xs, [w.detach() for w in weights], list(biases), 1))

    @staticmethod
    def backward(ctx, *dys):
        n, cin = ctx.n, ctx.cin
        big, *weights = ctx.saved_tensors
        B, H, W, _ = (int(v) for v in big.shape)
        dys = [torch.zeros(B, H, W, int(w.shape[0]), dtype=torch.float32, device=big.device) if d is None else d for d, w in zip(dys, weights)]
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[2:2 + n])
        need_db = any(h and g for h, g in zip(ctx.has_bias, ctx.needs_input_grad[2 + n:2 + 2 * n]))
        dbig = torch.empty_like(big) if need_dx else None
        xs = [big[..., i * cin:(i + 1) * cin] for i in range(n)]
        _, dws, dbs = thin_conv3x3_backward_batched(
            xs, dys, [w.detach() for w in weights], 1, need_dx=need_dx, need_dw=need_dw, need_db=need_db, dx_like=dbig,
            dw_outs=[grad_slots.claim(w) for w in weights] if need_dw else None,
            db_outs=[grad_slots.claim(b) if h else None for b, h in zip(ctx.bias_refs, ctx.has_bias)] if need_db else None)
        none = [None] * n
        dbs = none if dbs is None else [g if h else None for g, h in zip(dbs, ctx.has_bias)]
        return (dbig, None, *(dws or none), *dbs)


def sliced_thin_conv2d(big, convs):
    """[conv(big[..., i * cin:(i + 1) * cin]) for i, conv in enumerate(convs)] for thin 3x3 layers (``thin_conv_eligible`` on the
    slices); see _SlicedThinConv2dNHWC."""
    n = len(convs)
    return _SlicedThinConv2dNHWC.apply(big.contiguous(), n, *[c.weight for c in convs], *[c.bias for c in convs])


class _CatParams(torch.autograd.Function):
    """torch.cat(params, 0) whose backward hands every parameter its slice of the gradient through ONE multi-tensor copy, into the
    parameters' gradient slots of the flat buckets where they are free (autograd's own cat backward is a narrow + accumulate per
    parameter: 36 launches for the CenterHead's first-layer weights)."""
    @staticmethod
    def forward(ctx, *params):
        ctx.refs = params
        ctx.sizes = [int(p.shape[0]) for p in params]
        return torch.cat([p.detach() for p in params], 0)

    @staticmethod
    def backward(ctx, g):
        srcs = list(g.contiguous().split(ctx.sizes, 0))
        dsts = []
        for p, s_ in zip(ctx.refs, srcs):
            slot = grad_slots.claim(p)
            dsts.append(slot if slot is not None else torch.empty_like(s_))
        torch._foreach_copy_(dsts, srcs)
        return tuple(d if need else None for d, need in zip(dsts, ctx.needs_input_grad))


def cat_params(params):
    return _CatParams.apply(*params)


def multi_thin_conv2d(xs, convs):
    """[conv(x) for x, conv in zip(xs, convs)] for nn.Conv2d layers that satisfy ``thin_conv_eligible``; see _MultiThinConv2dNHWC."""
    n = len(convs)
    return _MultiThinConv2dNHWC.apply(n, *[x.contiguous() for x in xs], *[c.weight for c in convs], *[c.bias for c in convs])


def multi_conv2d(x, weights):
    """[conv2d(x, w, stride 1, pad 1) for w in weights] for 3x3 weights of one shape; see _MultiConv2dNHWC."""
    return _MultiConv2dNHWC.apply(_pad4(x), *weights)


def conv2d(x, weight, bias=None, stride=1, pad=0, dil=1):
    """``F.conv2d`` on NHWC float32 CUDA tensors through the HIP kernels, differentiable in ``x``, ``weight``, ``bias``.
    ``x`` may carry zero padding channels beyond ``weight.shape[1]``; a channel count that is not a multiple of 4 (the
    174-channel maps of the BSM head) is zero-padded here, because every kernel reads pixels in 16-byte pieces."""
    return _Conv2dNHWC.apply(_pad4(x), weight, bias, int(stride), int(pad), int(dil))


class _ConvTranspose2dNHWC(torch.autograd.Function):
    """nn.ConvTranspose2d with kernel == stride (the SECONDFPN deblocks): every input pixel owns a k x k output patch."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        assert int(weight.shape[2]) == int(weight.shape[3]) == stride
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        cp = int(x.shape[-1])
        return _packed_call(weight, ('deconv', stride, cp), lambda w: PackedConv(w, stride=stride, transposed=True, cin_pad=cp), x)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        k = ctx.stride
        dy = dy.contiguous()
        cin, cout = int(weight.shape[0]), int(weight.shape[1])
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dX[i, j, ci] = sum_{dy, dx, co} dY[i k + dy, j k + dx, co] W[ci, co, dy, dx]: a stride-k convolution of dY
            # whose OIHW weight is W as stored
            dyp = dy if int(dy.shape[-1]) % 4 == 0 else torch.nn.functional.pad(dy, (0, 4 - int(dy.shape[-1]) % 4))
            cp = int(dyp.shape[-1])
            dx = _packed_call(weight, ('deconv_dgrad', k, cp), lambda w: PackedConv(w, stride=k, cin_pad=cp, pad_out=True), dyp)
            if int(dx.shape[-1]) != int(x.shape[-1]):
                dx = torch.nn.functional.pad(dx, (0, int(x.shape[-1]) - int(dx.shape[-1])))
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(dy, x, k, k, 0, 1, cin=cout, cout=cin, out=grad_slots.claim(weight))      # roles swapped: OIHW = [cin, cout, k, k]
        return dx, dw, None


def conv_transpose2d(x, weight, stride):
    """``F.conv_transpose2d`` (kernel == stride, no bias) on NHWC float32 CUDA tensors, differentiable."""
    return _ConvTranspose2dNHWC.apply(_pad4(x), weight, int(stride))
