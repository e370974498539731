"""Thin host-side wrappers over the C ABI for the convolution family and the small layers.

Tensors here are torch CUDA tensors used purely as device buffers: activations are NHWC float32
``[B, H, W, C]`` (channel-last, the layout the gfx950 kernels want), every call enqueues on torch's
current stream.  No arithmetic happens in PyTorch on this path.
"""
import ctypes

import torch

from . import _lib, pack_cache
from ._lib import ConvDesc, CONV_NORMAL, CONV_DECONV, CONV_NCHW_OUT, CONV_GROUP_PLANES  # noqa: F401


def _st(t):
    return _lib.stream_handle(t.device)


# When a list, every wrapper appends (kernel name, algorithmic flops, start event, end event) recorded
# on the launch stream (bench.py's roofline pass).  None = no instrumentation.
PROFILE = None
# Pick algorithm / tile / split-K per (layer, input shape) by timing the candidates once, outside graph capture.
# False (SGV3D_NO_AUTOTUNE=1): a fixed rule instead -- the same choices on every run and every rank, hence
# bitwise reproducible results (measured choices can differ between runs with timing noise, which moves the
# results by fp32 rounding since the algorithms sum in different orders); ~10 % slower.
AUTOTUNE = not __import__("os").environ.get("SGV3D_NO_AUTOTUNE")
# True: conv records carry the layer shape in their name (tools/layer_report.py)
PROFILE_DETAIL = False
# False: the autotuner never proposes split-K (experiments; SGV3D_NO_SPLITK=1)
import os as _os
# Number of concurrent copies of a candidate the autotuner times (SGV3D_TUNE_STREAMS, default 1 = an isolated launch).
# With several frames in flight (pipeline.FramePipeline) a layer shares the chip with other frames' kernels: the
# throughput-optimal tile / split differs from the latency-optimal one an isolated launch finds (a split-K or small-tile
# choice that fills idle CUs in isolation only adds work when the CUs are busy anyway).  N > 1 launches the candidate on
# N streams at once and compares the time for all of them to finish.
TUNE_STREAMS = max(1, int(_os.environ.get("SGV3D_TUNE_STREAMS", "1")))
TUNE_VERBOSE = bool(_os.environ.get("SGV3D_TUNE_VERBOSE"))      # print every candidate's time as the first-call measurement goes
# launches per stream and repetitions of a candidate's timing under load (SGV3D_TUNE_ROUNDS / SGV3D_TUNE_REPEATS; the committed
# tune DBs are measured with 8 x 4 -- tools/make_tune_db.sh -- so that near-ties are not decided by noise)
TUNE_ROUNDS = max(1, int(_os.environ.get("SGV3D_TUNE_ROUNDS", "3")))
TUNE_REPEATS = max(1, int(_os.environ.get("SGV3D_TUNE_REPEATS", "2")))
SPLIT_K = not _os.environ.get("SGV3D_NO_SPLITK")
# False: 3x3 / stride-1 layers never use the Winograd F(2x2,3x3) kernel (SGV3D_NO_WINOGRAD=1)
WINOGRAD = not _os.environ.get("SGV3D_NO_WINOGRAD")
# False: CenterHead branches run as two kernels with the hidden maps in HBM (SGV3D_NO_FUSED_HEAD=1)
FUSED_HEAD = not _os.environ.get("SGV3D_NO_FUSED_HEAD")
# fp32 CenterHead branches (SGV3D_HEAD_PATH): 2 (default) = the fused F(4x4) kernel (csrc/head_wino4.hip: 532 us at the cfg-2
# launch); 0 = the fused F(2x2) kernel (conv_wino.hip: 976 us); 1 = first layers as one convolution (its algorithm measured per
# load: F(4x4) in channel chunks) + the final-conv kernel (executes as few MFMAs as path 2 but moves 3.9 GB of M / hidden maps per
# frame: 1264 vs 1089 us per call against path 0 with three in flight); "auto" (None here): all three timed under load at the
# first call.
HEAD_PATH = {"0": 0, "1": 1, "2": 2, "auto": None}.get(_os.environ.get("SGV3D_HEAD_PATH", ""), 2)
# True (SGV3D_BF16=1 or set before the first forward): every convolution multiplies through the bf16 MFMA variant of the
# implicit-GEMM kernel (operands rounded to bf16 on their way into LDS, fp32 accumulation and epilogue, fp32 tensors in
# HBM) -- the compute dtype BASELINE cfg-3 / cfg-5 name.  Winograd and the fused head kernel are fp32-only and are not
# used in this mode (a bf16 direct convolution runs at 16x the fp32 MFMA rate, so the 2.25x saving no longer matters).
MFMA_BF16 = bool(_os.environ.get("SGV3D_BF16"))
# bf16 mode only: the convolution chains of the ResNets (image backbone, BEV trunk) keep their activations as bf16
# tensors in HBM -- the 1x1 / strided layers at the large resolutions are HBM-bound, so halving the bytes is worth more
# than any MFMA tuning there (a 256 -> 1024 1x1 with residual at 68x120x4: 99 -> 55 us).  The stem conv writes bf16, the
# blocks read and write bf16 (epilogue in fp32, one rounding), the necks read bf16 and write fp32.
# SGV3D_NO_BF16_ACTIVATIONS=1 keeps fp32 tensors everywhere (round 1 behaviour).
BF16_ACTIVATIONS = not _os.environ.get("SGV3D_NO_BF16_ACTIVATIONS")
# bf16 compute mode in TRAINING (MFMA_BF16 with BF16_ACTIVATIONS off: f32 tensors, bf16 products): also the weight gradients
# run on the bf16 matrix cores (conv_wgrad_bf16_kernel); 0: they stay on the f32 MFMA kernel
TRAIN_BF16_WGRAD = _os.environ.get("SGV3D_TRAIN_BF16_WGRAD", "1") != "0"
# ... and 3x3 / stride-1 layers may use the all-taps form (conv_wgrad3x3_bf16.hip: a workgroup owns a 64 x 64 tile for all nine taps;
# the batched CenterHead launch always, single layers where the first-call measurement / the tune DB says so); 0: per-tap kernel only
WGRAD_BF16_ALLTAPS = _os.environ.get("SGV3D_WGRAD_BF16_ALLTAPS", "1") != "0"
# 1x1 layers with unpadded channel counts read the OIHW weight tensor itself as their packed weights (diagnostic: 0 packs a copy)
ALIAS_1X1_WEIGHTS = _os.environ.get("SGV3D_ALIAS_1X1_WEIGHTS", "1") != "0"
# True (SGV3D_F32X3=1): the implicit-GEMM layers compute float32-accurate products on the bf16 matrix cores -- every
# operand is split exactly into three bf16 terms and six partial products are accumulated in f32 (csrc/conv_igemm.hip,
# SPLIT3): the error of a product is one f32 rounding, the MFMA time 192 instead of 512 cycles per 16 k.  Winograd
# layers keep competing in the first-call measurement.  Opt-in: the default path multiplies on the f32 MFMA.
# SGV3D_F32X3=auto ("auto" here): the f32x3 tiles compete with the f32-MFMA tiles and Winograd in the first-call
# measurement of every layer (host-side tile ids 11..14 = f32x3 of tiles 1..4); long-K layers pick it, small-K ones do not.
MFMA_F32X3 = {"": False, "0": False, "auto": "auto"}.get(_os.environ.get("SGV3D_F32X3", ""), True)
# (tile, split-K) decisions by layer signature.  SGV3D_TUNE_CACHE=<file> loads them at import and
# save_tune_db() writes them back, so that a profiled run replays the choices of an earlier run
# instead of timing candidates again (keeps rocprofv3 per-kernel averages free of tuning launches).
TUNE_DB = {}
TUNE_STATS = {"measured": 0, "from_db": 0}     # per process: layer signatures timed here / answered from a tune DB
_COMMITTED_SIGS = set()      # signatures that came from tune/gfx950_*.json: honoured on gfx950 devices only
_TUNE_SIDE_STREAMS = []


def _tune_db_path():
    import os
    return os.environ.get("SGV3D_TUNE_CACHE")


def load_tune_db(path=None):
    import json
    import os
    path = path or _tune_db_path()
    if path and os.path.exists(path):
        with open(path) as f:
            loaded = {k: tuple(v) for k, v in json.load(f).items()}
        TUNE_DB.update(loaded)
        _COMMITTED_SIGS.difference_update(loaded)       # (a local measurement overrides a committed one: no longer arch-bound)
    return len(TUNE_DB)


def save_tune_db(path=None):
    import json
    path = path or _tune_db_path()
    if path:
        with open(path, "w") as f:
            json.dump({k: list(v) for k, v in TUNE_DB.items()}, f, indent=0, sort_keys=True)


def load_default_tune_dbs():
    """tune/gfx950_*.json at the repository root: the (algorithm, tile, split-K) choices measured on an MI355X for the
    BASELINE configurations, committed so that a run neither spends its first forward on candidate timing nor moves by
    near-tie picks from run to run.  A layer signature that is not in there is measured as before (autotune is the
    fallback).  SGV3D_NO_TUNE_DB=1 ignores the committed files, SGV3D_TUNE_SKIP=<name>[,<name>] only the named ones;
    SGV3D_TUNE_CACHE=<file> is loaded on top and is the file save_tune_db() writes."""
    import glob
    import os
    if os.environ.get("SGV3D_NO_TUNE_DB"):
        return 0
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tune")
    skip = {x for x in os.environ.get("SGV3D_TUNE_SKIP", "").split(",") if x}      # file names to leave out (re-measuring one DB)
    n = 0
    for f in sorted(glob.glob(os.path.join(root, "gfx950_*.json"))):
        if os.path.basename(f) in skip:
            continue
        with open(f) as fh:
            sigs = list(__import__("json").load(fh))
        n += load_tune_db(f) and 1
        _COMMITTED_SIGS.update(sigs)
    return n


_ARCH_IS_GFX950 = {}


def _is_gfx950(device):
    idx = torch.device(device).index or 0
    if idx not in _ARCH_IS_GFX950:
        _ARCH_IS_GFX950[idx] = "gfx950" in str(getattr(torch.cuda.get_device_properties(idx), "gcnArchName", ""))
    return _ARCH_IS_GFX950[idx]


load_default_tune_dbs()
load_tune_db()
TILE_NAMES = {1: "128x128", 2: "128x64", 3: "64x128", 4: "64x64", 5: "wino", 6: "wino_resident", 7: "patch_bf16", 8: "wino_half",
              9: "wino4", 10: "wino4", 15: "wino4",                         # F(4x4,3x3) with the 64x64 / 64x128 / 32x128 GEMM tile
              11: "128x128", 12: "128x64", 13: "64x128", 14: "64x64",      # 11..14: f32x3 of tiles 1..4 (host-side ids)
              21: "128x128", 22: "128x64", 23: "64x128", 24: "64x64",      # 21..24: tiles 1..4 walked m-tile first (SGV3D_TILE_MFIRST)
              31: "dw_bf16", 32: "dw_bf16", 33: "dw_bf16", 34: "dw_bf16", 35: "dw_bf16",   # bf16 direct-weight kernel (SGV3D_TILE_DW_*)
              36: "dw_bf16", 37: "dw_bf16",                                                # ... requests two k-chunks ahead (*_DEEP)
              38: "dw_bf16", 39: "dw_bf16",                                                # ... 64 pixels x 128 channels (39: two chunks ahead)
              40: "wino4_resident",    # F(4x4,3x3) with the transformed input resident in LDS (sgv3d_conv3x3_f4res_forward)
              47: "wino4",
              50: "wino4_x3", 51: "wino4_x3", 52: "wino4_x3", 53: "wino4_x3", 54: "wino4_x3",    # F(4x4) with the f32x3 position GEMM
              55: "wino4_x3", 56: "wino4_x3", 57: "wino4_x3", 58: "wino4_x3", 59: "wino4_x3",    # (csrc/gemm_x3_grouped.hip), by tile shape
              60: "pw_x3", 61: "pw_x3", 62: "pw_x3", 64: "pw_x3", 65: "pw_x3", 66: "pw_x3",      # pointwise f32x3 (csrc/conv_pw_x3.hip)
              70: "pw_x3", 71: "pw_x3", 72: "pw_x3", 74: "pw_x3", 75: "pw_x3", 76: "pw_x3",      # ... walked m-tile first
              80: "pw_x3", 81: "pw_x3", 82: "pw_x3", 90: "pw_x3", 91: "pw_x3", 92: "pw_x3",      # ... 256 channels per workgroup
              44: "64x64", 45: "64x64",    # the 64x64 tile at five workgroups per CU (SGV3D_TILE_OCC5); 45: walked m-tile first
              46: "wino4"}                 # F(4x4,3x3) in three launches with the five-per-CU 64x64 GEMM tile
MFIRST = _os.environ.get("SGV3D_MFIRST", "1") != "0"
TILE_WINO = 5       # host-side algorithm id: sgv3d_conv2d_winograd_forward instead of the implicit GEMM
TILE_WINO_RES = 6   # = SGV3D_WINOGRAD_RESIDENT: its patch-resident variant (cin <= 96, many cout tiles)
TILE_PATCH = 7      # bf16 mode: the LDS-resident-patch 3x3 kernel (sgv3d_conv3x3_patch_bf16_forward)
TILE_WINO_HALF = 8  # = SGV3D_WINOGRAD_HALF: 64 tiles x 32 channels per workgroup, positions split over wave pairs (2 workgroups / CU)
TILE_WINO4 = 9      # Winograd F(4x4,3x3) in three launches (sgv3d_conv2d_winograd4_forward), GEMM tile 64x64; 10: 64x128
TILE_WINO4_WIDE = 10
TILE_WINO4_NARROW = 15   # ... with the 32x128 GEMM tile: rows per position padded to 32 instead of 64 (336 tiles -> 352, 84 -> 96)
TILE_WINO4_OCC = 46      # ... with the five-workgroups-per-CU form of the 64x64 GEMM tile (SGV3D_TILE_64x64 | SGV3D_TILE_OCC5)
TILE_WINO4_G48 = 47      # ... with the grouped GEMM on v_mfma_f32_16x16x4_f32, 48 x 64 tiles (SGV3D_TILE_48x64): rows padded to 48 (336 -> 336)
# ... with the position GEMM on the bf16 matrix cores, f32-accurate ("f32x3": every operand split exactly into three bf16 terms by its
# PRODUCER -- the weight packer, the input transform --, six partial products accumulated in f32; csrc/gemm_x3_grouped.hip).  Host ids
# 50 + v: v % 5 = m-tile of {48, 64, 96, 112, 128} rows, v >= 5: 160 instead of 128 columns per workgroup.
WINO4_X3_TILES = tuple(range(50, 60))
X3_TILE_ROWS = {50 + v: (48, 64, 96, 112, 128)[v % 5] for v in range(10)}
X3_TILE_COLS = {50 + v: 160 if v >= 5 else 128 for v in range(10)}
WINO4_TILES = (TILE_WINO4, TILE_WINO4_WIDE, TILE_WINO4_NARROW, TILE_WINO4_OCC, TILE_WINO4_G48) + WINO4_X3_TILES
WINO4_G48 = _os.environ.get("SGV3D_WINO4_G48", "1") != "0"     # 0: never a candidate
# 0: the f32x3 position GEMM is never a candidate -- every product of the f32 path on the f32 MFMA (bench.py's native_f32_value)
WINO4_X3 = _os.environ.get("SGV3D_WINO4_X3", "1") != "0"
# Implicit-GEMM layers (1x1, strided 3x3 / 1x1, patchify: at most 32 taps, cin % 32 == 0) with f32-accurate products on the bf16 matrix
# cores (csrc/conv_pw_x3.hip: weights split into three
# bf16 terms by the packer, activations on their way into LDS).  Host ids 60 + v / 70 + v (m-tile first): v & 3 = {0: 32, 1: 64, 2: 128}
# pixels per workgroup, v & 4: 64 instead of 128 channels.  0: never a candidate.
PW_X3 = _os.environ.get("SGV3D_PW_X3", "1") != "0"
PW_X3_TILES = (60, 61, 62, 64, 65, 66, 70, 71, 72, 74, 75, 76, 80, 81, 82, 90, 91, 92)    # 8x / 9x: 256 channels per workgroup (8 waves)
PW_X3_DIMS = {t: (32 << ((t % 10) & 3), 256 if t >= 80 else 64 if (t % 10) & 4 else 128) for t in PW_X3_TILES}
# F(4x4,3x3) in ONE launch with V = B^T d B of a 16x16 block resident in LDS (csrc/head_wino4.hip: conv_f4res_kernel): 3x3 /
# stride 1 / pad 1 layers with 64 input channels (ResNet layer 1) or 64 output channels (the CenterHead's shared layer), f32
TILE_F4RES = 40
F4RES = _os.environ.get("SGV3D_F4RES", "1") != "0"     # 0: never a candidate
# f32 implicit GEMM, 64x64 tile with five workgroups per CU (32 KB of swizzled LDS, one register stage; csrc/conv_igemm.hip:
# OCC): for the small-K layers; host ids 44 / 45 (= m-tile first) -> SGV3D_TILE_64x64 | SGV3D_TILE_OCC5 [| SGV3D_TILE_MFIRST]
OCC5_TILES = (44, 45)
OCC5 = _os.environ.get("SGV3D_OCC5", "1") != "0"       # 0: never candidates
WINO4 = _os.environ.get("SGV3D_WINO4", "1") != "0"     # 0: F(4x4) is never a candidate
WINO4_MIN_CHANNELS = 128                              # candidates only where cin and cout are at least this
WINO_HALF = _os.environ.get("SGV3D_WINO_HALF", "1") != "0"
PATCH_BF16 = _os.environ.get("SGV3D_PATCH_BF16", "1") != "0"
# = SGV3D_TILE_DW_*: bf16 mode, bf16 tensors in and out: the direct-weight implicit GEMM (sgv3d_conv_dw_bf16_forward), pixels x
# channels per workgroup 64x256 / 128x128 / 256x64 (64 pixels per wave) and 128x256 / 256x128 (128 pixels per wave)
# 36 / 37 = SGV3D_TILE_DW_64x256_DEEP / 128x128_DEEP: rows and fragments requested two k-chunks ahead (launches of about one
# workgroup per CU, where nothing else hides the memory round trips); no split-K
# 38 = SGV3D_TILE_DW_64x128: one 32-channel tile per wave -- twice the workgroups of 64x256 on small maps; 39: its *_DEEP form
DW_TILES = (31, 32, 33, 34, 35, 36, 37, 38, 39)
DW_DEEP_TILES = (36, 37, 39)
DW_DEEP = _os.environ.get("SGV3D_DW_DEEP", "1") != "0"   # 0: the *_DEEP tiles are never candidates
DW_NARROW = _os.environ.get("SGV3D_DW_NARROW", "1") != "0"   # 0: the 64x128 tile is never a candidate
DW_DEEP_MAX_WGS = int(_os.environ.get("SGV3D_DW_DEEP_MAX_WGS", "768"))   # the *_DEEP tiles are candidates for grids up to this many workgroups
DW_BF16 = _os.environ.get("SGV3D_DW_BF16", "1") != "0"   # 0: never a candidate
DW_SPLIT_K = _os.environ.get("SGV3D_DW_SPLITK", "1") != "0"   # 0: the direct-weight kernel is never split along k


class prof:
    """``with prof("kernel", flops, nbytes):`` brackets a launch with HIP events when PROFILE is active.  ``flops`` /
    ``nbytes``: the launch's ALGORITHMIC work -- 2 x MACs of the direct convolution; every tensor the layer's
    definition touches read or written exactly once.  PROFILE records are ``(name, flops, start, stop, nbytes, extra)``; ``extra``: None or
    ``{'symbol': the MFMA kernel's exact symbol, 'mfma_flops': the flops that kernel EXECUTES in this launch}``."""
    __slots__ = ("name", "flops", "nbytes", "extra", "e0")

    def __init__(self, name, flops=0.0, nbytes=0.0, extra=None):
        self.name, self.flops, self.nbytes, self.extra, self.e0 = name, flops, nbytes, extra, None

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.name, self.flops, self.e0, e1, self.nbytes, self.extra))
            self.e0 = None
        return False


def heuristic_tile(M, N):
    """Same cost model as pick_tile() in csrc/conv_igemm.hip."""
    best, best_cost = 1, None
    for t, (bm, bn, pen) in enumerate(((128, 128, 1.0), (128, 64, 1.06), (64, 128, 1.06), (64, 64, 1.18)), 1):
        tiles = -(-M // bm) * -(-N // bn)
        cost = -(-tiles // 256) * bm * bn * pen
        if best_cost is None or cost < best_cost:
            best, best_cost = t, cost
    return best


def pack_geometry(k, n):
    kp, np_ = ctypes.c_int(), ctypes.c_int()
    _lib.load().sgv3d_conv_pack_geometry(int(k), int(n), ctypes.byref(kp), ctypes.byref(np_))
    return kp.value, np_.value


def fold_bn(bn, conv_bias=None):
    """Eval-mode BatchNorm folded to per-channel (scale, shift):  y = scale * conv + shift."""
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        gamma = bn.weight.float() if bn.weight is not None else torch.ones_like(inv)
        beta = bn.bias.float() if bn.bias is not None else torch.zeros_like(inv)
        scale = gamma * inv
        shift = beta - bn.running_mean.float() * scale
        if conv_bias is not None:
            shift = shift + conv_bias.float() * scale
    return scale.contiguous(), shift.contiguous()


def activation_dtype(*channel_counts):
    """torch.bfloat16 when bf16 mode keeps activations as bf16 tensors in HBM (MFMA_BF16 and BF16_ACTIVATIONS, not the f32x3
    split) and every given channel count allows 16-byte bf16 rows (multiple of 8); else None (= float32 tensors)."""
    if not (MFMA_BF16 and BF16_ACTIVATIONS) or MFMA_F32X3:
        return None
    return torch.bfloat16 if all(int(c) % 8 == 0 for c in channel_counts) else None


def channel_align():
    """Granularity the layers that own their activation buffers pad channel counts to: 4 (16-byte f32 rows); 32 in
    bf16-activation mode, where a multiple of 32 input channels opens the chunk-major k order, bf16 tensors in HBM and the
    patch kernel (SGV3D's 87 / 174 / 348 / 696-channel BEV trunk -> 96 / 192 / 352 / 704; the extra channels are exact zeros)."""
    return 32 if activation_dtype() is not None else 4


def pad_channels(c, align=None):
    align = int(align or channel_align())
    return (int(c) + align - 1) // align * align


class PackedConv:
    """One convolution (or kernel==stride transposed convolution) with its weights repacked for the
    MFMA implicit-GEMM kernel and BN / bias folded into a per-channel scale & shift."""

    def __init__(self, weight, *, stride=1, pad=0, dil=1, scale=None, shift=None, relu=False,
                 transposed=False, cin_pad=None, device=None, tile=0, pad_out=False):
        w = weight.detach()
        device = device or w.device
        w = w.to(device=device, dtype=torch.float32).contiguous()
        # Channel counts that are not multiples of 4 (SGV3D's 87 / 174-channel BEV trunk) are padded:
        # inputs with zero weight columns (always), outputs with zero rows when the caller owns the
        # activation buffer (pad_out): the extra output channels come out exactly 0 and feed zero
        # weights downstream, so every pixel row stays 16-byte aligned for the kernels.
        odim = 1 if transposed else 0
        self.cout_real = int(w.shape[odim])
        out_align = 4 if pad_out is True else int(pad_out or 0)       # pad_out: True = 4, or the alignment itself
        if out_align and self.cout_real % out_align:
            extra = out_align - self.cout_real % out_align
            shape = list(w.shape)
            shape[odim] = extra
            w = torch.cat([w, w.new_zeros(shape)], odim).contiguous()
            if scale is not None:
                scale = torch.cat([scale.detach().float().to(device), torch.ones(extra, device=device)])
            if shift is not None:
                shift = torch.cat([shift.detach().float().to(device), torch.zeros(extra, device=device)])
        self.cin_real = int(w.shape[0] if transposed else w.shape[1])    # what the algorithmic flop count prices
        if cin_pad is None:
            cin_pad = (self.cin_real + 3) // 4 * 4
        if transposed:
            cin, cout, kh, kw = (int(s) for s in w.shape)
            assert kh == kw == stride, "only kernel == stride transposed convs (SECONDFPN deblocks)"
            self.ks = kh
            self.kh = self.kw = 1
            self.stride, self.pad, self.dil = 1, 0, 1
            gemm_n = cout * kh * kw
        else:
            cout, cin, kh, kw = (int(s) for s in w.shape)
            self.ks = 0
            self.kh, self.kw = kh, kw
            self.stride, self.pad, self.dil = int(stride), int(pad), int(dil)
            gemm_n = cout
        self.transposed = bool(transposed)
        self.cin = int(cin_pad) if cin_pad else cin
        assert self.cin >= cin and self.cin % 4 == 0, f"cin={cin} must be padded to a multiple of 4"
        self.cout = cout
        self.relu = bool(relu)
        self.tile = int(tile)
        k = self.cin if transposed else kh * kw * self.cin
        self.k_pad, self.cout_pad = pack_geometry(k, gemm_n)
        self.k_order = 1 if self.cin % 32 == 0 else 0      # channel-chunk-major k when possible
        self.scale = None if scale is None else scale.detach().to(device=device, dtype=torch.float32).contiguous()
        self.shift = None if shift is None else shift.detach().to(device=device, dtype=torch.float32).contiguous()
        # Every packed form of the weights (implicit GEMM, F(2x2) / F(4x4) Winograd, bf16 patch / direct-weight) is made on
        # first use: a training step packs the current weights of every layer twice (forward, data gradient) and only
        # needs the form its kernel choice reads.
        self._w = self._w_wino = None
        self._entry = None          # pack_cache._Entry when this object is kept across training steps (sgv3d_amd/pack_cache.py)
        self.device = device
        # Winograd F(2x2,3x3) covers the layer
        self.wino_ok = bool(WINOGRAD and not transposed and kh == 3 and kw == 3 and self.stride == 1 and self.dil == 1
                            and self.pad == 1 and self.cin % 8 == 0)
        self._keep = w  # the pack kernels read it asynchronously
        self._tile_cache = {}
        # bf16 mode: fragment-ordered bf16 weights for the patch kernel, packed on first use
        self.w_patch = None
        self.w_bf16 = None
        self.patch_ok = (not transposed and kh == 3 and kw == 3 and self.stride == 1 and self.dil == 1
                         and self.pad == 1 and self.cin % 32 == 0 and self.cout % 8 == 0 and self.cin == cin)

    @property
    def w(self):
        """Weights packed for the implicit-GEMM kernels (sgv3d_conv_pack_weight), [cout_pad, k_pad] f32."""
        if self._w is None:
            src = self._keep
            odim = 1 if self.transposed else 0
            cout, cin = int(src.shape[odim]), int(src.shape[1 - odim])
            if (ALIAS_1X1_WEIGHTS and not self.transposed and self.kh == 1 and self.kw == 1 and cout == self.cout_pad and cin == self.k_pad
                    and self.cin == cin and src.is_contiguous() and src.data_ptr() % 16 == 0):
                # a 1x1 layer whose channel counts need no padding: OIHW [cout, cin, 1, 1] IS the packed layout [cout_pad, k_pad] (one tap:
                # both k orders are the identity) -- no pack launch (a training step repacks every layer's weights, forward and data
                # gradient, at ~5 us per launch)
                self._w = src.view(self.cout_pad, self.k_pad)
                if self._entry is not None and src.data_ptr() != self._entry.param.data_ptr():
                    # kept across steps and NOT a view of the parameter itself (the rotated copy of a data gradient): refreshed like a
                    # packed form, in place
                    self._entry.register('w', self._w, lambda pc: pc.w)
                return self._w
            self._w = torch.empty(self.cout_pad, self.k_pad, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                rc = _lib.load().sgv3d_conv_pack_weight(src.data_ptr(), cout, cin, int(src.shape[2]), int(src.shape[3]), self.cin,
                                                       1 if self.transposed else 0, self.k_order, self._w.data_ptr(),
                                                       self.k_pad, self.cout_pad, _st(src))
            _lib.check(rc, "sgv3d_conv_pack_weight")
            if self._entry is not None:
                self._entry.register('w', self._w, lambda pc: pc.w)
        return self._w

    @property
    def w_wino(self):
        """Winograd F(2x2,3x3) weights (sgv3d_conv_winograd_pack_weight), None when the layer is not covered."""
        if self._entry is not None:
            raise pack_cache.NotAPermutation('w_wino')     # a transformed form: this layer packs per call (pack_cache._Entry.call)
        if self._w_wino is None and self.wino_ok:
            lib = _lib.load()
            src = self._keep
            self._w_wino = torch.empty(lib.sgv3d_conv_winograd_weight_floats(self.cout, self.cin), dtype=torch.float32,
                                       device=self.device)
            with torch.cuda.device(self.device):
                rc = lib.sgv3d_conv_winograd_pack_weight(src.data_ptr(), self.cout, int(src.shape[1]), self.cin,
                                                         self._w_wino.data_ptr(), _st(src))
            _lib.check(rc, "sgv3d_conv_winograd_pack_weight")
        return self._w_wino

    def _bf16_weights(self):
        """bf16 copy of the packed implicit-GEMM weights (bf16-activation launches), made on first use."""
        if self.w_bf16 is None:
            self.w_bf16 = torch.empty(self.cout_pad, self.k_pad, dtype=torch.bfloat16, device=self.device)
            with torch.cuda.device(self.device):
                rc = _lib.load().sgv3d_conv_weight_to_bf16(self.w.data_ptr(), self.k_pad, self.cout_pad, self.w_bf16.data_ptr(), _st(self.w))
            _lib.check(rc, "sgv3d_conv_weight_to_bf16")
            if self._entry is not None:
                self._entry.register('w_bf16', self.w_bf16, lambda pc: pc._bf16_weights())
        return self.w_bf16

    def wino4_ok(self, d=None, gate=None):
        """F(4x4,3x3) covers this layer (and launch): 3x3 / stride 1 / pad 1, f32, NHWC output, no gate, many channels."""
        is3x3 = (not self.transposed and self.kh == 3 and self.kw == 3 and self.stride == 1 and self.pad == self.dil)   # dilated too
        # enough channels for the transforms to be worth it -- or very many output channels on few input ones (the 64 -> 36 x 64
        # first layers of the CenterHead branches: the input transform is shared by all of them)
        wide = min(self.cin, self.cout) >= WINO4_MIN_CHANNELS or (self.cin >= 64 and self.cout >= 1024)
        ok = (WINO4 and WINOGRAD and is3x3 and not MFMA_BF16 and self.cin % 32 == 0 and self.cout % 4 == 0 and wide and gate is None)
        if ok and d is not None:
            ok = ((d.mode == CONV_NORMAL and d.y_ld % 4 == 0 and d.y_coff % 4 == 0) or
                  (d.mode == CONV_GROUP_PLANES and d.deconv_ks % 4 == 0)) and d.x_ld % 4 == 0 and d.x_coff % 4 == 0 and d.res_ld % 4 == 0
        return ok

    def f4res_ok(self, d=None, gate=None, io=0):
        """The resident F(4x4) kernel covers this layer (and launch): f32, 3x3 / stride 1 / pad 1 / dilation 1, NHWC in and
        out, no gate, and 64 input channels with 64 n output channels or 64 n input channels with 64 output channels."""
        ok = (F4RES and WINO4 and WINOGRAD and not MFMA_BF16 and not MFMA_F32X3 and not self.transposed and self.kh == 3
              and self.kw == 3 and self.stride == 1 and self.pad == 1 and self.dil == 1 and gate is None and io == 0
              and self.cin % 64 == 0 and self.cout % 64 == 0 and (self.cin == 64 or self.cout == 64))
        if ok and d is not None:
            ok = (d.mode == CONV_NORMAL and d.x_ld % 4 == 0 and d.x_coff % 4 == 0 and d.y_ld % 4 == 0 and d.y_coff % 4 == 0
                  and d.res_ld % 4 == 0)
        return ok

    def _f4res_weights(self):
        """U = G g G^T in the fragment order conv_f4res_kernel streams (sgv3d_conv3x3_f4res_pack_weight), made on first use."""
        if self._entry is not None:
            raise pack_cache.NotAPermutation('w_f4res')     # a transformed form: this layer packs per call (pack_cache._Entry.call)
        if getattr(self, 'w_f4res', None) is None:
            lib = _lib.load()
            w = self._keep                                           # [cout, cin_real, 3, 3] f32 on the device
            self.w_f4res = torch.empty(int(lib.sgv3d_conv3x3_f4res_weight_floats(self.cout, self.cin)), dtype=torch.float32,
                                       device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv3x3_f4res_pack_weight(w.data_ptr(), self.cout, int(w.shape[1]), self.cin,
                                                         self.w_f4res.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv3x3_f4res_pack_weight")
        return self.w_f4res

    def _wino4_weights(self):
        """U[p] = (G g G^T)[i][j] for the 36 positions of F(4x4,3x3), each a packed 1x1 weight block of the implicit-GEMM
        kernel (36 x cout_pad x k_pad floats), made on first use by one kernel (sgv3d_conv_winograd4_pack_weight)."""
        if self._entry is not None:
            raise pack_cache.NotAPermutation('w_wino4')     # a transformed form: this layer packs per call (pack_cache._Entry.call)
        if getattr(self, 'w_wino4', None) is None:
            lib = _lib.load()
            w = self._keep                                           # [cout, cin_real, 3, 3] f32 on the device
            k_pad, cout_pad = pack_geometry(self.cin, self.cout)
            packed = torch.empty(36, cout_pad, k_pad, dtype=torch.float32, device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv_winograd4_pack_weight(w.data_ptr(), self.cout, int(w.shape[1]), k_pad, cout_pad,
                                                          packed.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv_winograd4_pack_weight")
            self.w_wino4, self.wino4_geom = packed, (k_pad, cout_pad)
        return self.w_wino4

    def pw_x3_ok(self, d=None, gate=None, io=0):
        """The implicit-GEMM f32x3 kernel (csrc/conv_pw_x3.hip) covers this layer (and launch): a convolution with at most 32 taps and
        cin % 32 == 0 -- the 1x1 layers, the strided 3x3 / 1x1 layers between the stages, the patchify layers of the necks -- or, with
        tap-major weights, at most 64 taps and cin % 4 == 0 (the 7x7 stems); f32 tensors, NHWC output, no gate."""
        taps = self.kh * self.kw
        shape_ok = ((self.k_order == 1 and self.cin % 32 == 0 and self.cin >= 64 and taps <= 32) or        # channel-chunk-major k
                    (self.k_order == 0 and self.cin % 4 == 0 and taps <= 64 and taps * self.cin >= 128))   # tap-major k: the 7x7 stems
        ok = (PW_X3 and not MFMA_BF16 and not MFMA_F32X3 and not self.transposed and gate is None and io == 0 and shape_ok
              and self.cout % 4 == 0)
        if ok and d is not None:
            ok = (d.mode == CONV_NORMAL and d.x_ld % 4 == 0 and d.x_coff % 4 == 0 and d.y_ld % 4 == 0 and d.y_coff % 4 == 0
                  and d.res_ld % 4 == 0)
        return ok

    def _pw_x3_weights(self):
        """The weights as three bf16 planes per element in fragment order ([cout_pad / 16][kh kw cin / 32][3][512], cout_pad = cout
        rounded up to 32), made on first use (sgv3d_conv_pack_weight_x3)."""
        if self._entry is not None:
            raise pack_cache.NotAPermutation('w_pw_x3')     # a transformed form: this layer packs per call (pack_cache._Entry.call)
        if getattr(self, 'w_pw_x3', None) is None:
            lib = _lib.load()
            w = self._keep                                           # [cout, cin_real, kh, kw] f32 on the device
            cout_pad = (self.cout + 31) // 32 * 32
            k_pad = (self.kh * self.kw * self.cin + 31) // 32 * 32
            packed = torch.empty(cout_pad // 16, k_pad // 32, 3, 512, dtype=torch.bfloat16, device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv_pack_weight_x3(w.data_ptr(), self.cout, int(w.shape[1]), self.kh, self.kw, self.cin, cout_pad,
                                                   packed.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv_pack_weight_x3")
            self.w_pw_x3, self.pw_x3_cout_pad = packed, cout_pad
        return self.w_pw_x3

    def _wino4_x3_weights(self):
        """U[p] of F(4x4,3x3) as three bf16 planes per element ([36][cout_pad][cin / 32][3][32], cout_pad = cout rounded up to 32) for
        the f32x3 position GEMM, made on first use by one kernel (sgv3d_conv_winograd4_pack_weight_x3)."""
        if self._entry is not None:
            raise pack_cache.NotAPermutation('w_wino4_x3')     # a transformed form: this layer packs per call (pack_cache._Entry.call)
        if getattr(self, 'w_wino4_x3', None) is None:
            lib = _lib.load()
            w = self._keep                                           # [cout, cin_real, 3, 3] f32 on the device
            cout_pad = (self.cout + 31) // 32 * 32
            packed = torch.empty(36, cout_pad, self.cin // 32, 3, 32, dtype=torch.bfloat16, device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv_winograd4_pack_weight_x3(w.data_ptr(), self.cout, int(w.shape[1]), self.cin, cout_pad,
                                                             packed.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv_winograd4_pack_weight_x3")
            self.w_wino4_x3, self.wino4_x3_cout_pad = packed, cout_pad
        return self.w_wino4_x3

    def _dw_weights(self):
        """Fragment-ordered bf16 weights of the direct-weight kernel (sgv3d_conv_dw_bf16_pack_weight), made on first use.
        Transposed convolution (kernel == stride): a 1x1 layer with ks * ks * cout outputs ordered (dy, dx, co)."""
        if getattr(self, 'w_dw', None) is None:
            lib = _lib.load()
            w = self._keep                                           # [cout, cin_real, kh, kw] f32 on the device ([cin_real, cout, ks, ks] transposed)
            n = self.cout
            if self.transposed:
                w = w.permute(2, 3, 1, 0).reshape(self.ks * self.ks * self.cout, int(w.shape[0]), 1, 1).contiguous()
                n = self.ks * self.ks * self.cout
            self.w_dw = torch.empty(lib.sgv3d_conv_dw_bf16_weight_bytes(n, self.cin, self.kh, self.kw), dtype=torch.uint8, device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv_dw_bf16_pack_weight(w.data_ptr(), n, int(w.shape[1]), self.cin, self.kh, self.kw,
                                                        self.w_dw.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv_dw_bf16_pack_weight")
            self._keep_dw = w
            if self._entry is not None:
                self._entry.register('w_dw', self.w_dw, lambda pc: pc._dw_weights())
        return self.w_dw

    def _dw_eligible(self, d, gate=None, io=0):
        return (MFMA_BF16 and not MFMA_F32X3 and DW_BF16 and io == 3 and d.mode in (CONV_NORMAL, CONV_DECONV)
                and gate is None and self.cin % 32 == 0 and self.cout % 8 == 0 and d.x_ld % 8 == 0 and d.x_coff % 8 == 0
                and d.y_ld % 8 == 0 and d.y_coff % 8 == 0 and d.res_ld % 8 == 0)

    def _patch_weights(self):
        if self.w_patch is None:
            lib = _lib.load()
            w = self._keep
            self.w_patch = torch.empty(lib.sgv3d_conv3x3_patch_bf16_weight_bytes(self.cout, self.cin), dtype=torch.uint8,
                                       device=w.device)
            with torch.cuda.device(w.device):
                rc = lib.sgv3d_conv3x3_patch_bf16_pack_weight(w.data_ptr(), self.cout, self.cin, self.w_patch.data_ptr(), _st(w))
            _lib.check(rc, "sgv3d_conv3x3_patch_bf16_pack_weight")
            if self._entry is not None:
                self._entry.register('w_patch', self.w_patch, lambda pc: pc._patch_weights())
        return self.w_patch

    def out_hw(self, h, w):
        if self.transposed:
            return h * self.ks, w * self.ks
        oh = (h + 2 * self.pad - self.dil * (self.kh - 1) - 1) // self.stride + 1
        ow = (w + 2 * self.pad - self.dil * (self.kw - 1) - 1) // self.stride + 1
        return oh, ow

    def __call__(self, x, out=None, *, shift=None, **kw):
        """``shift``: a per-call replacement of the layer's per-channel shift (f32 [cout] on the device) -- for a layer whose
        bias depends on the input of THIS call (ASPP's pooled branch folded into the 1x1 convolution behind the concat)."""
        if shift is None:
            return self._call(x, out, **kw)
        assert shift.dtype == torch.float32 and shift.numel() == self.cout and shift.is_contiguous() and shift.device == x.device
        saved, self.shift = self.shift, shift
        try:
            return self._call(x, out, **kw)
        finally:
            self.shift = saved

    def _call(self, x, out=None, *, x_coff=0, y_coff=0, residual=None, gate=None, nchw_out=False, tile=None,
              group_planes=0, split_k=None, out_dtype=None):
        """x NHWC [B,H,W,x_ld]; reads channels [x_coff, x_coff+cin).  out NHWC [B,OH,OW,y_ld] written
        at channels [y_coff, y_coff+cout) (allocated [B,OH,OW,cout] if None).
        bf16 mode: ``x`` may be a bf16 tensor, and ``out_dtype=torch.bfloat16`` (or a bf16 ``out``) makes the layer write
        bf16 (then ``residual`` is bf16 too) -- NORMAL layout only."""
        B, H, W, x_ld = (int(s) for s in x.shape)
        assert x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
        if out is not None:
            out_dtype = out.dtype
        out_dtype = out_dtype or torch.float32
        io = (1 if x.dtype == torch.bfloat16 else 0) | (2 if out_dtype == torch.bfloat16 else 0)
        if io:
            if not MFMA_BF16 or MFMA_F32X3:
                raise _lib.SGV3DError("bf16 tensors are only handled in bf16 mode (hip_ops.MFMA_BF16)")
            if io & 2:
                assert not (nchw_out or group_planes or gate is not None), "bf16 output: NORMAL / transposed-conv layouts only"
                assert residual is None or residual.dtype == torch.bfloat16
            else:
                assert residual is None or residual.dtype == torch.float32
        oh, ow = self.out_hw(H, W)
        if group_planes:
            # [cout/g, B, OH, OW, g]: one NHWC map per group of g output channels
            assert self.cout % group_planes == 0 and not self.transposed and residual is None
            if out is None:
                out = torch.empty(self.cout // group_planes, B, oh, ow, group_planes, dtype=torch.float32, device=x.device)
            assert out.is_contiguous() and out.numel() == B * oh * ow * self.cout
        elif out is None:
            shape = (B, self.cout, oh, ow) if nchw_out else (B, oh, ow, self.cout)
            out = torch.empty(shape, dtype=out_dtype, device=x.device)
        else:
            want = (B, None, oh, ow) if nchw_out else (B, oh, ow, None)
            got = tuple(int(v) for v in out.shape)
            if len(got) != 4 or any(w is not None and w != g for w, g in zip(want, got)) or not out.is_contiguous():
                raise _lib.SGV3DError(f"conv output buffer {got} does not match {want}")
        y_ld = int(out.shape[1] if nchw_out else out.shape[-1])
        if group_planes:
            y_ld, y_coff = self.cout, 0
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.cin = B, H, W, self.cin
        d.out_h, d.out_w, d.cout = oh, ow, self.cout
        d.kh, d.kw, d.stride, d.pad, d.dil = self.kh, self.kw, self.stride, self.pad, self.dil
        d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, int(x_coff), y_ld, int(y_coff)
        d.res_ld = int(residual.shape[-1]) if residual is not None else 0
        d.relu = 1 if self.relu else 0
        d.mode = CONV_DECONV if self.transposed else (CONV_NCHW_OUT if nchw_out else CONV_NORMAL)
        d.deconv_ks = self.ks
        if group_planes:
            d.mode, d.deconv_ks = CONV_GROUP_PLANES, int(group_planes)
        d.k_pad, d.cout_pad = self.k_pad, self.cout_pad
        d.x_nchw = 0
        d.k_order = self.k_order
        lib = _lib.load()
        gemm_m = B * (H * W if self.transposed else oh * ow)
        gemm_n = self.cout * (self.ks * self.ks if self.transposed else 1)
        nkt = self.k_pad // 32
        t = int(self.tile if tile is None else tile)
        sk = int(split_k) if split_k else 0
        if t == 0 or sk == 0:
            key = (B, H, W, t, sk, MFMA_BF16, MFMA_F32X3, d.mode, residual is not None, gate is not None, io)
            choice = self._tile_cache.get(key)
            if choice is None:
                sig = (f"{self.cout}x{self.cin}k{self.kh}x{self.kw}s{self.stride}p{self.pad}d{self.dil}"
                       f"ks{self.ks}|{B}x{H}x{W}|m{d.mode}r{int(residual is not None)}g{int(gate is not None)}|{t}.{sk}"
                       + ("|bf16" if MFMA_BF16 else "|f32x3" if MFMA_F32X3 is True else "|x3auto" if MFMA_F32X3 else "")
                       + (f"|io{io}" if io else "") + f"|ts{TUNE_STREAMS}")    # choices are per frames-in-flight load
                choice = self._db_choice(sig, d, x, gate, gemm_m, gemm_n, nkt, t, sk, io)
                if choice is not None:
                    self._tile_cache[key] = choice
                    TUNE_STATS["from_db"] += 1
                elif AUTOTUNE and not torch.cuda.is_current_stream_capturing():
                    TUNE_STATS["measured"] += 1
                    choice = self._autotune(lib, d, x, residual, gate, out, gemm_m, gemm_n, nkt, t, sk, io)
                    self._tile_cache[key] = choice
                    TUNE_DB[sig] = choice
                    _COMMITTED_SIGS.discard(sig)        # measured on THIS device
                else:
                    choice = self._rule(t, sk, d, gemm_m, gemm_n, gate)
            t, sk = choice
        d.tile, d.split_k = t, sk
        # ALGORITHMIC flops: real channel counts, not the zero-padded ones the kernel multiplies
        real_n = self.cout_real * (self.ks * self.ks if self.transposed else 1)
        flops = 2.0 * gemm_m * real_n * (self.cin_real * self.kh * self.kw)
        x3 = 10 < t < 20 or (MFMA_F32X3 is True and t < TILE_WINO)
        name = ("conv_" if t in (TILE_WINO, TILE_WINO_RES, TILE_PATCH, TILE_WINO_HALF, TILE_F4RES) + WINO4_TILES + DW_TILES + PW_X3_TILES else
                ("conv_igemm_bf16_" if MFMA_BF16 else "conv_igemm_f32x3_" if x3 else "conv_igemm_")) + TILE_NAMES[t]
        plain = t not in (TILE_WINO, TILE_WINO_RES, TILE_PATCH, TILE_WINO_HALF, TILE_F4RES) + WINO4_TILES + DW_TILES + PW_X3_TILES
        if plain and self.k_order == 0:
            name += "_tapmajor"        # the <.., false> instantiation (cin % 32 != 0: stems), a different kernel symbol
        elif plain and not MFMA_BF16 and not x3:
            # the label names the INSTANTIATION launch_t (csrc/conv_igemm.hip) picks, so that a per-symbol rocprof / PMC
            # summary can be attached to exactly this kernel (bench.py load_traffic): pointwise and five-per-CU forms
            pw = (self.kh == 1 and self.kw == 1 and self.stride == 1 and self.pad == 0 and not self.transposed)
            occ = t in OCC5_TILES
            if pw and (occ or self.cin >= 128) and (occ or TILE_NAMES[t] in ("64x64", "64x128")):
                name += "_pw"
            if occ:
                name += "_occ5"
        # algorithmic bytes: input map (the channels this layer reads), weights, output (+ residual), each once
        # (element sizes as the tensors are stored: bf16 activations in HBM count 2 bytes)
        nbytes = (float(x.element_size()) * B * H * W * self.cin_real
                  + (2.0 if MFMA_BF16 else 4.0) * self.cout_real * self.cin_real * self.kh * self.kw * (self.ks * self.ks if self.transposed else 1)
                  + float(out.element_size()) * gemm_m * real_n
                  + (float(residual.element_size()) * gemm_m * real_n if residual is not None else 0.0))
        if PROFILE_DETAIL:
            name += (f"|{B}x{H}x{W}x{self.cin}->{self.cout} k{self.kh if not self.transposed else -self.ks} "
                     f"s{self.stride} d{self.dil} splitk{sk}" + (" mfirst" if 20 < t < 30 or t == 45 else "") + (" occ5" if t in OCC5_TILES else ""))
        if io:
            name = name.replace("conv_igemm_bf16_", "conv_igemm_bf16io_")
        extra = None
        if PROFILE is not None and not MFMA_BF16 and not x3:
            b = lambda v: "true" if v else "false"
            if plain:
                wtm, wtn = (int(v) // 64 for v in TILE_NAMES[t].split("x"))
                extra = {"symbol": f"conv_igemm_kernel<{wtm}, {wtn}, {b(self.k_order != 0)}, false, {b(name.endswith(('_pw', '_pw_occ5')))}, "
                                   f"false, {b(t in OCC5_TILES)}>",
                         "mfma_flops": 2.0 * gemm_m * self.cout * self.cin * self.kh * self.kw * (self.ks * self.ks if self.transposed else 1)}
            elif t in PW_X3_TILES:
                bm, bn = PW_X3_DIMS[t]
                pw1 = self.kh == 1 and self.kw == 1 and self.stride == 1 and self.pad == 0
                kk = (self.cin * self.kh * self.kw + 31) // 32 * 32
                extra = {"symbol": f"conv_pw_x3_kernel<{bm // 16}, {bn // 32}, {2 if self.k_order == 0 else 0 if pw1 else 1}>",
                         "mfma_flops": 2.0 * gemm_m * self.cout * kk, "bf16_mfma_flops": 6 * 2.0 * gemm_m * self.cout * kk}
            elif t in WINO4_TILES:
                # the grouped GEMM of the three-launch F(4x4) path: 36 positions x rows (tiles padded to the GEMM's m-tile)
                dil = max(1, self.dil)
                tiles = B * dil * dil * -(-(-(-oh // dil)) // 4) * -(-(-(-ow // dil)) // 4)
                g = X3_TILE_ROWS[t] if t in WINO4_X3_TILES else 32 if t == TILE_WINO4_NARROW else 48 if t == TILE_WINO4_G48 else 64
                rows = -(-tiles // g) * g
                if t in WINO4_X3_TILES:
                    # six bf16 partial products per f32 product: the kernel's roofline is the bf16 MFMA peak over these flops
                    extra = {"symbol": f"gemm_x3_grouped_kernel<{g // 16}, {X3_TILE_COLS[t] // 32}>",
                             "mfma_flops": 2.0 * 36 * rows * self.cin * self.cout, "bf16_mfma_flops": 6 * 2.0 * 36 * rows * self.cin * self.cout}
                else:
                    extra = {"symbol": {TILE_WINO4: "conv_igemm_kernel<1, 1, true, false, true, false, false>",
                                        TILE_WINO4_WIDE: "conv_igemm_kernel<1, 2, true, false, true, false, false>",
                                        TILE_WINO4_NARROW: "conv_igemm_kernel<1, 1, true, false, true, true, false>",
                                        TILE_WINO4_OCC: "conv_igemm_kernel<1, 1, true, false, true, false, true>",
                                        TILE_WINO4_G48: "gemm16_grouped_kernel<3>"}[t],
                             "mfma_flops": 2.0 * 36 * rows * self.cin * self.cout}
        with torch.cuda.device(x.device), prof(name, flops, nbytes, extra):
            rc = self._launch(lib, d, x, residual, gate, out, io)
        _lib.check(rc, "sgv3d_conv2d_forward")
        return out

    def _rule(self, t, sk, d, gemm_m, gemm_n, gate=None):
        """Deterministic choice without measurement: Winograd for the layers it covers once the map has
        enough tiles to occupy the chip (split over channel steps to reach ~2 workgroups per CU), else the
        implicit-GEMM tile of the cost model; explicit tile / split arguments win."""
        if t == 0 and self.wino_ok and WINOGRAD and not MFMA_BF16 and d.out_h * d.out_w * d.batch >= 1024:
            wgs = d.batch * -(-d.out_h // 16) * -(-d.out_w // 16) * -(-gemm_n // 64)
            split = 1
            if not sk and SPLIT_K:
                for cand in (2, 3, 4, 6, 8):
                    if wgs * cand <= 512 and self.cin // 8 // cand >= 4:
                        split = cand
            return TILE_WINO, sk or split
        if t == 0 and self._patch_eligible(d, gate) and d.out_h * d.out_w * d.batch >= 4096:
            wgs = d.batch * -(-d.out_h // 16) * -(-d.out_w // 32) * -(-gemm_n // 64)
            split = 1
            if not sk and SPLIT_K:
                for cand in (2, 3, 4, 6, 8):
                    if wgs * cand <= 512 and self.cin // 32 // cand >= 2:
                        split = cand
            return TILE_PATCH, sk or split
        return (t or heuristic_tile(gemm_m, gemm_n)), (sk or 1)

    def _patch_eligible(self, d, gate=None):
        return (MFMA_BF16 and not MFMA_F32X3 and PATCH_BF16 and self.patch_ok and d.mode == CONV_NORMAL and gate is None
                and d.x_ld % 8 == 0 and d.x_coff % 8 == 0 and d.y_ld % 8 == 0 and d.y_coff % 8 == 0
                and (d.res_ld % 8 == 0))

    def _launch(self, lib, d, x, residual, gate, out, io=0):
        ws, nws = None, 0
        if d.split_k > 1 and d.tile not in DW_TILES:
            nws = lib.sgv3d_conv2d_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        if d.tile == TILE_PATCH:
            if not self._patch_eligible(d, gate):
                raise _lib.SGV3DError("the bf16 patch kernel covers 3x3 / stride 1 / pad 1 layers with cin % 32 == 0 in bf16 mode")
            return lib.sgv3d_conv3x3_patch_bf16_forward(d.batch, d.in_h, d.in_w, self.cin, self.cout, d.x_ld, d.x_coff, d.y_ld,
                                                        d.y_coff, d.res_ld, d.relu, x.data_ptr(), self._patch_weights().data_ptr(),
                                                        _lib.ptr(self.scale), _lib.ptr(self.shift), _lib.ptr(residual),
                                                        out.data_ptr(), int(io), int(d.split_k), _lib.ptr(ws), nws, _st(x))
        if d.tile in DW_TILES:
            if not self._dw_eligible(d, gate, io) or (d.split_k > 1 and d.mode != CONV_NORMAL):
                raise _lib.SGV3DError("the bf16 direct-weight kernel covers layers with cin % 32 == 0, cout % 8 == 0 and bf16 tensors in "
                                      "and out (NHWC), no gate; split-K in NORMAL mode only")
            if d.split_k > 1:
                nws = lib.sgv3d_conv_dw_bf16_workspace_bytes(ctypes.byref(d))
                ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
                return lib.sgv3d_conv_dw_bf16_forward_splitk(ctypes.byref(d), x.data_ptr(), self._dw_weights().data_ptr(),
                                                             _lib.ptr(self.scale), _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(),
                                                             ws.data_ptr(), nws, _st(x))
            return lib.sgv3d_conv_dw_bf16_forward(ctypes.byref(d), x.data_ptr(), self._dw_weights().data_ptr(), _lib.ptr(self.scale),
                                                  _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(), _st(x))
        if d.tile == TILE_F4RES:
            if not self.f4res_ok(d, gate, io) or d.split_k > 1 or x.dtype != torch.float32:
                raise _lib.SGV3DError("the resident F(4x4) kernel covers f32 3x3 / stride 1 / pad 1 layers with 64 input channels "
                                      "(cout % 64 == 0) or 64 output channels (cin % 64 == 0), NHWC output, no gate, no split-K")
            return lib.sgv3d_conv3x3_f4res_forward(ctypes.byref(d), x.data_ptr(), self._f4res_weights().data_ptr(), _lib.ptr(self.scale),
                                                   _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(), _st(x))
        if d.tile in PW_X3_TILES:
            if not self.pw_x3_ok(d, gate, io) or x.dtype != torch.float32:
                raise _lib.SGV3DError("the implicit-GEMM f32x3 kernel covers f32 layers with at most 32 taps, cin % 32 == 0 (>= 64), cout % 4 == 0, "
                                      "NHWC output, no gate")
            u = self._pw_x3_weights()
            host_tile, cp = d.tile, d.cout_pad
            # SGV3D_TILE_X3 | variant [| SGV3D_TILE_MFIRST]: 6x / 7x: variant = t % 10; 8x / 9x: 8 + t % 10 (256-channel tiles)
            d.tile = 64 | ((host_tile % 10) + (8 if host_tile >= 80 else 0)) | (16 if host_tile // 10 in (7, 9) else 0)
            d.cout_pad = self.pw_x3_cout_pad
            try:
                return lib.sgv3d_conv2d_x3_forward(ctypes.byref(d), x.data_ptr(), u.data_ptr(), _lib.ptr(self.scale), _lib.ptr(self.shift),
                                                   _lib.ptr(residual), out.data_ptr(), _lib.ptr(ws), nws, _st(x))
            finally:
                d.tile, d.cout_pad = host_tile, cp
        if d.tile in WINO4_TILES:
            if not self.wino4_ok(d, gate) or d.split_k > 1:
                raise _lib.SGV3DError("F(4x4) Winograd covers f32 3x3 / stride 1 / pad 1 layers with cin % 32 == 0, cout % 4 == 0, "
                                      "NHWC output, no gate, no split-K")
            host_tile, kp, cp = d.tile, d.k_pad, d.cout_pad
            if host_tile in WINO4_X3_TILES:
                u = self._wino4_x3_weights()
                d.tile = 64 | (host_tile - 50)                     # SGV3D_TILE_X3 | variant
                d.k_pad, d.cout_pad = self.cin, self.wino4_x3_cout_pad
            else:
                u = self._wino4_weights()
                d.tile = {TILE_WINO4: 4, TILE_WINO4_WIDE: 3, TILE_WINO4_NARROW: 9, TILE_WINO4_OCC: 4 | 32,
                          TILE_WINO4_G48: 10}[host_tile]      # SGV3D_TILE_64x64 / 64x128 / 32x128 / 64x64 | OCC5 / 48x64
                d.k_pad, d.cout_pad = self.wino4_geom
            try:
                nws4 = lib.sgv3d_conv2d_winograd4_workspace_bytes(ctypes.byref(d))
                ws4 = torch.empty(nws4, dtype=torch.uint8, device=x.device)
                return lib.sgv3d_conv2d_winograd4_forward(ctypes.byref(d), x.data_ptr(), u.data_ptr(), _lib.ptr(self.scale),
                                                          _lib.ptr(self.shift), _lib.ptr(residual), out.data_ptr(), ws4.data_ptr(),
                                                          nws4, _st(x))
            finally:
                d.tile, d.k_pad, d.cout_pad = host_tile, kp, cp
        if d.tile in (TILE_WINO, TILE_WINO_RES, TILE_WINO_HALF):
            if not self.wino_ok:
                raise _lib.SGV3DError("this layer has no Winograd weights (needs 3x3 / stride 1 / pad 1 / cin % 8 == 0)")
            return lib.sgv3d_conv2d_winograd_forward(ctypes.byref(d), x.data_ptr(), self.w_wino.data_ptr(),
                                                     _lib.ptr(self.scale), _lib.ptr(self.shift), _lib.ptr(residual),
                                                     _lib.ptr(gate), out.data_ptr(), _lib.ptr(ws), nws, _st(x))
        x3 = 10 < d.tile < 20 or (MFMA_F32X3 is True)
        host_tile = d.tile
        if host_tile in OCC5_TILES:
            if MFMA_BF16 or x3 or io or self.k_order != 1:
                raise _lib.SGV3DError("the five-per-CU 64x64 tile is f32 only and needs channel-chunk-major weights (cin % 32 == 0)")
            d.tile = 4 | 32 | (16 if host_tile == 45 else 0)      # SGV3D_TILE_64x64 | SGV3D_TILE_OCC5 [| SGV3D_TILE_MFIRST]
        elif host_tile > 20:
            d.tile = (host_tile - 20) | 16          # SGV3D_TILE_MFIRST
        elif host_tile > 10:
            d.tile = host_tile - 10
        if io:
            try:
                return lib.sgv3d_conv2d_forward_bf16io(ctypes.byref(d), x.data_ptr(), self._bf16_weights().data_ptr(), _lib.ptr(self.scale),
                                                       _lib.ptr(self.shift), _lib.ptr(residual), _lib.ptr(gate), out.data_ptr(),
                                                       _lib.ptr(ws), nws, _st(x), int(io))
            finally:
                d.tile = host_tile
        fwd = (lib.sgv3d_conv2d_forward_bf16 if MFMA_BF16 else
               lib.sgv3d_conv2d_forward_f32x3 if x3 else lib.sgv3d_conv2d_forward)
        try:
            return fwd(ctypes.byref(d), x.data_ptr(), self.w.data_ptr(), _lib.ptr(self.scale),
                       _lib.ptr(self.shift), _lib.ptr(residual), _lib.ptr(gate), out.data_ptr(),
                       _lib.ptr(ws), nws, _st(x))
        finally:
            d.tile = host_tile

    def _time_under_load(self, lib, d, x, residual, gate, out, rounds=None, io=0):
        """Time for TUNE_STREAMS concurrent copies of the launch, ``rounds`` back to back on every stream (all copies
        write the same values to ``out``; split-K workspaces are per launch)."""
        global _TUNE_SIDE_STREAMS
        cur = torch.cuda.current_stream(x.device)
        if len(_TUNE_SIDE_STREAMS) < TUNE_STREAMS:
            _TUNE_SIDE_STREAMS = [torch.cuda.Stream(device=x.device) for _ in range(TUNE_STREAMS)]
        best = None
        rounds = rounds or TUNE_ROUNDS
        for _ in range(TUNE_REPEATS):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for s in _TUNE_SIDE_STREAMS[:TUNE_STREAMS]:
                s.wait_event(e0)
                with torch.cuda.stream(s):
                    for _r in range(rounds):
                        self._launch(lib, d, x, residual, gate, out, io)
                cur.wait_stream(s)
            e1.record(cur)
            e1.synchronize()
            dt = e0.elapsed_time(e1)
            best = dt if best is None else min(best, dt)
        return best

    def _candidates(self, d, gate, gemm_m, gemm_n, nkt, fixed_tile, fixed_split, io=0):
        """The (tile, (split-K, ...)) pairs this layer may run with on this input / output layout under the module's current
        switches (WINOGRAD, WINO4, WINO_HALF, PATCH_BF16, DW_*, SPLIT_K, MFIRST, MFMA_*): what the first-call measurement times,
        and what a tune-DB entry is checked against before it is trusted (``_db_choice``)."""
        tiles = (1, 2, 3, 4)
        if MFIRST and d.mode in (CONV_NORMAL, CONV_NCHW_OUT) and gemm_n > 64:
            tiles += tuple(t + 20 for t in (1, 2, 3, 4) if gemm_n > (128 if t in (1, 3) else 64))   # more than one channel tile
        if OCC5 and self.k_order == 1 and not MFMA_BF16 and not MFMA_F32X3 and io == 0:
            tiles += (44,)
            if MFIRST and d.mode in (CONV_NORMAL, CONV_NCHW_OUT) and gemm_n > 64:
                tiles += (45,)
        if MFMA_F32X3 == "auto" and not MFMA_BF16:
            tiles += (11, 12, 13, 14)
        if self.wino_ok and WINOGRAD and not MFMA_BF16:
            tiles += (TILE_WINO,)
            if WINO_HALF:
                tiles += (TILE_WINO_HALF,)
            if self.cin <= 96 and self.cout >= 128:
                tiles += (TILE_WINO_RES,)

        if self.wino4_ok(d, gate):            # (also the dilated 3x3 layers, which the F(2x2) kernels do not cover)
            tiles += ((TILE_WINO4, TILE_WINO4_WIDE) + ((TILE_WINO4_NARROW,) if self.cin >= 128 else ()) + ((TILE_WINO4_OCC,) if OCC5 else ())
                      + ((TILE_WINO4_G48,) if WINO4_G48 and self.k_order == 1 else ()))
            if WINO4_X3 and not MFMA_F32X3:
                tiles += self._x3_tiles(d)
        if self.f4res_ok(d, gate, io):
            tiles += (TILE_F4RES,)
        if self.pw_x3_ok(d, gate, io):
            # shapes that leave the chip neither empty nor padded: at least ~128 workgroups, m-tiles no larger than the map needs
            for t in PW_X3_TILES:
                bm, bn = PW_X3_DIMS[t]
                wgs = -(-gemm_m // bm) * -(-gemm_n // bn)
                mfirst = t // 10 in (7, 9)
                if bn == 256 and gemm_n < 256:
                    continue
                if (wgs >= 96 or (SPLIT_K and nkt >= 16)) and (bn >= 128 or gemm_n <= 64 or wgs < 1024) and (not mfirst or (MFIRST and gemm_n > bn)):
                    tiles += (t,)
        if self._patch_eligible(d, gate):
            tiles += (TILE_PATCH,)
        if self._dw_eligible(d, gate, io):
            deep = self.kh * self.kw * self.cin >= 512              # enough k for the 128-pixel wave tiles to pay
            tiles += ((31,) + ((34,) if deep else ()) if gemm_n > 128 else (32,) + ((35,) if deep else ()) if gemm_n > 64 else (32, 33))
            if DW_DEEP and self.kh * self.kw * self.cin >= 256:     # few workgroups per CU: the two-chunks-ahead form of the 64-pixel tiles
                if gemm_n > 128 and -(-gemm_m // 64) * -(-gemm_n // 256) <= DW_DEEP_MAX_WGS:
                    tiles += (36,)
                    if DW_NARROW and -(-gemm_m // 64) * -(-gemm_n // 256) <= 384:
                        tiles += (38, 39)
                elif 64 < gemm_n <= 128 and -(-gemm_m // 128) * -(-gemm_n // 128) <= DW_DEEP_MAX_WGS:
                    tiles += (37,)
        if fixed_tile:
            tiles = (fixed_tile,)
        dims = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64), 5: (256, 64), 6: (256, 64),
                11: (128, 128), 12: (128, 64), 13: (64, 128), 14: (64, 64),
                21: (128, 128), 22: (128, 64), 23: (64, 128), 24: (64, 64), 44: (64, 64), 45: (64, 64)}
        dims.update(PW_X3_DIMS)
        cands = []
        for t in tiles:
            bm, bn = dims.get(t, (512, 64))
            wgs = -(-gemm_m // bm) * -(-gemm_n // bn)
            nk = nkt
            if t in (TILE_WINO, TILE_WINO_HALF):
                nk = self.cin // 4      # k-steps of 8 channels; nk // s >= 8 keeps >= 4 steps per split
                wgs = d.batch * -(-d.out_h // 16) * -(-d.out_w // 16) * -(-gemm_n // (64 if t == TILE_WINO else 32))
            if t == TILE_PATCH:
                nk = self.cin // 32     # stages of 32 input channels; >= 2 per split
                wgs = d.batch * -(-d.out_h // 16) * -(-d.out_w // 32) * -(-gemm_n // 64)
            if t in DW_TILES:
                nk = -(-(self.kh * self.kw * (self.cin // 32)) // 2)      # chunks of 64 k; >= 4 per split
                bm, bn = {31: (64, 256), 32: (128, 128), 33: (256, 64), 34: (128, 256), 35: (256, 128), 36: (64, 256), 37: (128, 128), 38: (64, 128), 39: (64, 128)}[t]
                wgs = -(-gemm_m // bm) * -(-gemm_n // bn)
            if t in (TILE_WINO_RES, TILE_F4RES) or t in WINO4_TILES or t in DW_DEEP_TILES or (t in DW_TILES and (d.mode != CONV_NORMAL or not DW_SPLIT_K)):
                splits = (1,)
            elif t in DW_TILES:
                splits = (fixed_split,) if fixed_split else \
                    [1] + [s for s in (2, 3, 4, 6, 8) if SPLIT_K and nk // s >= 4 and wgs < 384 and wgs * s <= 1024]
            elif t == TILE_PATCH and not fixed_split and SPLIT_K:
                splits = [1] + [s for s in (2, 3, 4, 6, 8) if nk // s >= 2 and wgs * s <= 1024]
            elif fixed_split:
                splits = (fixed_split,)
            elif not SPLIT_K:
                splits = (1,)
            else:
                splits = [1] + [s for s in (2, 3, 4, 6, 8) if nk // s >= 4 and wgs < 2048 and wgs * s <= 6144]
            cands.append((t, tuple(splits)))
        return cands

    def _x3_tiles(self, d):
        """The shapes of the f32x3 position GEMM worth timing for this layer: m-tiles that waste at most 10 % of the rows as padding
        (the smallest waste always), 160 columns only where they cover cout with fewer idle columns than 128 do."""
        dil = max(1, int(d.dil))
        tiles = d.batch * dil * dil * -(-(-(-d.out_h // dil)) // 4) * -(-(-(-d.out_w // dil)) // 4)
        waste = {r: (-(-tiles // r) * r - tiles) / tiles for r in (48, 64, 96, 112, 128)}
        best = min(waste.values())
        ms = [i for i, r in enumerate((48, 64, 96, 112, 128)) if waste[r] <= max(best, 0.10) and (r <= 64 or tiles >= r)]
        idle = lambda c: -(-self.cout // c) * c - self.cout
        ws = [0] + ([5] if idle(160) < idle(128) else [])
        return tuple(50 + w + m for w in ws for m in ms)

    def _db_choice(self, sig, d, x, gate, gemm_m, gemm_n, nkt, fixed_tile, fixed_split, io=0):
        """The tune-DB entry of ``sig`` if it is one of the candidates this layer would be measured with right now, else None
        (and the stale entry is dropped, so the caller measures or applies the rule).  A DB is a record of measurements, not
        an override: the kill switches (SGV3D_NO_WINOGRAD, SGV3D_WINO4=0, SGV3D_DW_BF16=0, SGV3D_NO_SPLITK, ...), an
        input / output layout the recorded kernel does not cover (channel offsets / strides are not part of the
        signature) and SGV3D_NO_AUTOTUNE (the documented fixed rule) all win over it, and the committed ``tune/gfx950_*``
        entries only apply on a gfx950 device."""
        if not AUTOTUNE:
            return None
        choice = TUNE_DB.get(sig)
        if choice is None:
            return None
        if sig in _COMMITTED_SIGS and not _is_gfx950(x.device):
            return None
        t, sk = int(choice[0]), int(choice[1])
        for ct, splits in self._candidates(d, gate, gemm_m, gemm_n, nkt, fixed_tile, fixed_split, io):
            if ct == t and sk in splits:
                return (t, sk)
        del TUNE_DB[sig]
        _COMMITTED_SIGS.discard(sig)
        return None

    def _autotune(self, lib, d, x, residual, gate, out, gemm_m, gemm_n, nkt, fixed_tile, fixed_split, io=0):
        """Time the candidate (tile, split-K) pairs on the real buffers and keep the fastest.  Results do
        not depend on the tile shape (every output element sums k in the same order); split-K changes
        the association of the k sum (partials added in fixed order), still deterministic."""
        cands = self._candidates(d, gate, gemm_m, gemm_n, nkt, fixed_tile, fixed_split, io)
        best, best_t = (cands[0][0], fixed_split or 1), None
        with torch.cuda.device(x.device):
            for t, splits in cands:
                for sk in splits:
                    d.tile, d.split_k = t, sk
                    _lib.check(self._launch(lib, d, x, residual, gate, out, io), "sgv3d_conv2d_forward")   # warm
                    if TUNE_STREAMS > 1:
                        dt = self._time_under_load(lib, d, x, residual, gate, out, io=io)
                    else:
                        nrep = max(4, TUNE_ROUNDS * TUNE_REPEATS // 2)
                        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nrep + 1)]
                        evs[0].record()
                        for r in range(nrep):
                            self._launch(lib, d, x, residual, gate, out, io)
                            evs[r + 1].record()
                        evs[-1].synchronize()
                        dt = min(evs[r].elapsed_time(evs[r + 1]) for r in range(nrep))
                    if TUNE_VERBOSE:
                        print(f"[tune] {self.cout}x{self.cin}k{self.kh} {d.batch}x{d.in_h}x{d.in_w}: tile {t} split {sk}: {dt * 1e3:.1f} us", flush=True)
                    if best_t is None or dt < best_t:
                        best, best_t = (t, sk), dt
        return best


PAIR_BF16 = _os.environ.get("SGV3D_PAIR_BF16", "1") != "0"   # 0: conv2 + conv3 of a bottleneck are always two launches


# Independent pieces of ONE forward as parallel branches of the captured hipGraph -- BUILT, MEASURED, OFF BY DEFAULT
# (SGV3D_PARALLEL_BRANCHES=1 turns it on).  A batch-1 frame is a chain of ~130 launches, and the narrow ones -- the strided
# shortcut of a residual stage, the four SECONDFPN levels, the 27-feature gate MLPs, the pooled ASPP branch -- leave most CUs
# idle while the next launch waits for them although it does not need their result.  Inside a stream capture ``run_parallel``
# puts every piece on a forked stream and joins them, so the graph gets parallel branches; outside a capture the pieces run in
# order on the current stream.  Every piece writes its own buffer or channel slice: results are bitwise the sequential ones.
# Measured on cfg-2 (round 4, one call, same box): the ~30 fork / join pairs per frame cost more than the overlap returns --
# one frame in flight 169.3 -> 160.7 frames/s, three in flight 209.0 -> 163.3, the harness step 153.3 -> 147.8: every
# cross-stream edge of a hipGraph is a barrier packet plus a semaphore between hardware queues, tens of microseconds each,
# where the kernels it lets overlap are 10-60 us long.
PARALLEL_BRANCHES = _os.environ.get("SGV3D_PARALLEL_BRANCHES", "0") == "1"
_BRANCH_POOL = {}
_BRANCH_TOP = {}


_OWNED_STREAMS = {}       # device index -> cuda_stream handles handed out by distinct_stream() and still meant to be distinct


def distinct_stream(device, avoid=()):
    """A ``torch.cuda.Stream`` whose underlying HIP stream is none of ``avoid`` (streams), not the current stream and none that
    this function handed out before on the device.  ``torch.cuda.Stream()`` draws round-robin from a pool of 32 per device: the
    33rd object IS the first one's stream again, and a fork of a stream capture onto the capturing stream itself (or onto another
    branch of the same capture) is not the graph the code meant -- after enough pipelines / graphed forwards in one process that
    happened (a crash inside hipStreamEndCapture in the GPU suite, round 6)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    taken = _OWNED_STREAMS.setdefault(key, set())
    banned = {st.cuda_stream for st in avoid if st is not None} | {torch.cuda.current_stream(dev).cuda_stream} | taken
    for _ in range(64):
        st = torch.cuda.Stream(device=dev)
        if st.cuda_stream not in banned:
            if len(taken) < 24:                     # (beyond that the pool cannot keep everything distinct: best effort)
                taken.add(st.cuda_stream)
            return st
    return torch.cuda.Stream(device=dev)


def release_stream(st):
    """Give a stream of ``distinct_stream`` back (its owner is gone)."""
    for taken in _OWNED_STREAMS.values():
        taken.discard(st.cuda_stream)


def run_parallel(device, thunks):
    """``[t() for t in thunks]``; inside a stream capture the thunks are forked branches of the graph (the first one stays on
    the capturing stream).  A call from inside a forked branch runs its thunks in sequence."""
    thunks = list(thunks)
    if len(thunks) <= 1 or not PARALLEL_BRANCHES or not torch.cuda.is_current_stream_capturing():
        return [t() for t in thunks]
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _BRANCH_POOL.setdefault(key, [])
    top = _BRANCH_TOP.get(key, 0)
    if top > 0:
        # already inside a forked branch: in sequence (forks nested three deep -- MSCThead's scales -> ASPP -> pooled branch --
        # crashed hipStreamEndCapture on ROCm 7.2)
        return [t() for t in thunks]
    need = top + len(thunks) - 1
    while len(pool) < need:
        pool.append(distinct_stream(dev, pool))
    mine = pool[top:need]
    _BRANCH_TOP[key] = need
    cur = torch.cuda.current_stream(dev)
    results = [None] * len(thunks)
    try:
        for st, (i, t) in zip(mine, list(enumerate(thunks))[1:]):
            st.wait_stream(cur)                                   # fork
            with torch.cuda.stream(st):
                results[i] = t()
        results[0] = thunks[0]()
    finally:
        for st in mine:
            cur.wait_stream(st)                                   # join (every forked stream must rejoin the capture)
        _BRANCH_TOP[key] = top
    return results


# A capture that wants to be cut into several hipGraphs (pipeline.GraphedForward) installs a callback here; the forward calls
# graph_split_point() where a cut is useful (after the image backbone's first stage).  None: the call does nothing.
_GRAPH_SPLIT_HOOK = None


def graph_split_point(tag):
    if _GRAPH_SPLIT_HOOK is not None:
        _GRAPH_SPLIT_HOOK(tag)


def switch_state():
    """Every module-level switch that decides WHICH kernels a forward launches, as one hashable value: part of the key of
    anything that records launches for replay (BEVHeight's per-signature hipGraph)."""
    g = globals()
    return tuple(g.get(k) for k in ("AUTOTUNE", "SPLIT_K", "WINOGRAD", "FUSED_HEAD", "HEAD_PATH", "MFMA_BF16", "BF16_ACTIVATIONS",
                                    "MFMA_F32X3", "MFIRST", "WINO4", "WINO_HALF", "PATCH_BF16", "DW_BF16", "DW_DEEP", "DW_NARROW",
                                    "DW_SPLIT_K", "DW_DEEP_MAX_WGS", "PAIR_BF16", "TUNE_STREAMS", "PARALLEL_BRANCHES", "F4RES", "OCC5", "WINO4_G48", "DCN_FUSED", "WINO4_X3", "PW_X3"))


def conv_pair_eligible(a, b, x, residual=None):
    """conv ``a`` (k x k, 256 outputs, ReLU) followed by the 1x1 conv ``b`` (cout % 256 == 0) between bf16 NHWC tensors: the pair
    the fused direct-weight kernel (sgv3d_conv_dw_bf16_pair_forward) covers."""
    return (MFMA_BF16 and not MFMA_F32X3 and DW_BF16 and PAIR_BF16 and x.dtype == torch.bfloat16 and not a.transposed and not b.transposed
            and a.cout == 256 and a.cin % 32 == 0 and int(x.shape[-1]) % 8 == 0 and b.cin == 256 and b.kh == 1 and b.kw == 1
            and b.stride == 1 and b.pad == 0 and b.cout % 256 == 0 and (residual is None or residual.dtype == torch.bfloat16))


def conv_pair_bf16(a, b, x, residual=None, out=None):
    """``b(a(x), residual=residual)`` for two PackedConvs in one launch (bf16 tensors); the map between them stays in LDS."""
    B, H, W, x_ld = (int(v) for v in x.shape)
    oh, ow = a.out_hw(H, W)
    if out is None:
        out = torch.empty(B, oh, ow, b.cout, dtype=torch.bfloat16, device=x.device)
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.cin = B, H, W, a.cin
    d.out_h, d.out_w, d.cout = oh, ow, a.cout
    d.kh, d.kw, d.stride, d.pad, d.dil = a.kh, a.kw, a.stride, a.pad, a.dil
    d.x_ld, d.x_coff, d.y_ld, d.y_coff = x_ld, 0, a.cout, 0
    d.relu, d.mode, d.split_k = (1 if a.relu else 0), CONV_NORMAL, 1
    flops = 2.0 * B * oh * ow * (a.cout_real * a.cin_real * a.kh * a.kw + b.cout_real * b.cin_real)
    name = "conv_dw_bf16_pair"
    if PROFILE_DETAIL:
        name += f"|{B}x{H}x{W}x{a.cin}->{a.cout} k{a.kh} s{a.stride} d{a.dil} ->{b.cout} k1"
    with torch.cuda.device(x.device), prof(name, flops):
        rc = _lib.load().sgv3d_conv_dw_bf16_pair_forward(
            ctypes.byref(d), x.data_ptr(), a._dw_weights().data_ptr(), _lib.ptr(a.scale), _lib.ptr(a.shift), b.cout,
            b._dw_weights().data_ptr(), _lib.ptr(b.scale), _lib.ptr(b.shift), _lib.ptr(residual),
            int(residual.shape[-1]) if residual is not None else 0, out.data_ptr(), int(out.shape[-1]), 0, 1 if b.relu else 0, _st(x))
    _lib.check(rc, "sgv3d_conv_dw_bf16_pair_forward")
    return out


def conv_pair_choice(a, b, x, residual=None):
    """True when the fused launch is faster than ``b(a(x))`` with the per-layer choices: measured once per pair and input shape
    under the load the pipeline runs (like the per-layer choices) and kept in the tune DB under a ``pair|`` signature."""
    if not conv_pair_eligible(a, b, x, residual):
        return False
    B, H, W, _ = (int(v) for v in x.shape)
    sig = (f"pair|{a.cout}x{a.cin}k{a.kh}x{a.kw}s{a.stride}p{a.pad}d{a.dil}->{b.cout}|{B}x{H}x{W}|r{int(residual is not None)}|bf16|ts{TUNE_STREAMS}")
    if not AUTOTUNE:                                         # the documented fixed rule: no recorded measurement is consulted
        return True
    if sig in TUNE_DB and not (sig in _COMMITTED_SIGS and not _is_gfx950(x.device)):
        return bool(TUNE_DB[sig][0])
    if torch.cuda.is_current_stream_capturing():
        return True
    act = torch.bfloat16
    two = lambda: b(a(x, out_dtype=act), residual=residual, out_dtype=act)
    one = lambda: conv_pair_bf16(a, b, x, residual)
    two(); one()
    t2 = time_callable(two, x.device)
    t1 = time_callable(one, x.device)
    TUNE_DB[sig] = [1 if t1 < t2 else 0, 0]
    _COMMITTED_SIGS.discard(sig)
    return t1 < t2


def time_callable(fn, device, rounds=None):
    """Milliseconds for TUNE_STREAMS concurrent copies of ``fn`` (``rounds`` calls back to back on every stream; one stream =
    isolated timing), best of TUNE_REPEATS -- how the per-layer candidates are timed, for whole sub-graphs (the CenterHead
    branch paths).  ``fn`` launches on torch's current stream."""
    global _TUNE_SIDE_STREAMS
    rounds = rounds or TUNE_ROUNDS
    cur = torch.cuda.current_stream(device)
    n = max(1, TUNE_STREAMS)
    if len(_TUNE_SIDE_STREAMS) < n:
        _TUNE_SIDE_STREAMS = [torch.cuda.Stream(device=device) for _ in range(n)]
    best = None
    for _ in range(TUNE_REPEATS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for s_ in _TUNE_SIDE_STREAMS[:n]:
            s_.wait_event(e0)
            with torch.cuda.stream(s_):
                for _r in range(rounds):
                    fn()
            cur.wait_stream(s_)
        e1.record(cur)
        e1.synchronize()
        dt = e0.elapsed_time(e1)
        best = dt if best is None else min(best, dt)
    return best


def maxpool3x3s2(x, out=None):
    B, H, W, C = (int(s) for s in x.shape)
    oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty(B, oh, ow, C, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), prof("maxpool3x3s2"):
        if x.dtype == torch.bfloat16:
            rc = _lib.load().sgv3d_maxpool3x3s2_bf16(B, H, W, C, x.data_ptr(), out.data_ptr(), _st(x))
        else:
            rc = _lib.load().sgv3d_maxpool3x3s2(B, H, W, C, x.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_maxpool3x3s2")
    return out


def nchw_to_nhwc(x, c_pad=None, out=None):
    B, C, H, W = (int(s) for s in x.shape)
    assert x.is_contiguous() and x.dtype == torch.float32
    c_pad = int(c_pad or C)
    if out is None:
        out = torch.empty(B, H, W, c_pad, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), prof("nchw_to_nhwc"):
        rc = _lib.load().sgv3d_nchw_to_nhwc(B, C, H, W, c_pad, x.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, channels=None, coff=0, out=None):
    B, H, W, ld = (int(s) for s in x.shape)
    C = int(channels or ld)
    if out is None:
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), prof("nhwc_to_nchw"):
        rc = _lib.load().sgv3d_nhwc_to_nchw(B, C, H, W, ld, int(coff), x.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_nhwc_to_nchw")
    return out


def global_avgpool(x, out=None):
    B, H, W, C = (int(s) for s in x.shape)
    if out is None:
        out = torch.empty(B, C, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.sgv3d_global_avgpool_workspace_bytes(B, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    fn = lib.sgv3d_global_avgpool_bf16 if x.dtype == torch.bfloat16 else lib.sgv3d_global_avgpool
    with torch.cuda.device(x.device), prof("global_avgpool"):
        rc = fn(B, H * W, C, C, x.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, _st(x))
    _lib.check(rc, "sgv3d_global_avgpool")
    return out


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def dense(x, w, scale=None, bias=None, act=ACT_NONE, out=None, run=None):
    """x [B,K], w [N,K] -> act(scale*(x @ w.T) + bias) [B,N].  ``run``: int32 device flag; 0 skips the launch's work and leaves
    ``out`` (then required) as it is (sgv3d_dense_gated)."""
    B, K = (int(s) for s in x.shape)
    N = int(w.shape[0])
    assert int(w.shape[1]) == K and x.is_contiguous() and w.is_contiguous()
    assert run is None or out is not None
    if out is None:
        out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (B, N) and out.dtype == torch.float32 and out.is_contiguous()
    with torch.cuda.device(x.device), prof("dense"):
        if run is not None:
            rc = _lib.load().sgv3d_dense_gated(B, K, N, x.data_ptr(), w.data_ptr(), _lib.ptr(scale), _lib.ptr(bias),
                                              int(act), out.data_ptr(), run.data_ptr(), _st(x))
        else:
            rc = _lib.load().sgv3d_dense(B, K, N, x.data_ptr(), w.data_ptr(), _lib.ptr(scale), _lib.ptr(bias),
                                        int(act), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_dense")
    return out


def broadcast_channels(v, out, y_coff=0):
    """v [B,C] written to out[b, :, :, y_coff:y_coff+C] for every pixel (out NHWC)."""
    B, H, W, ld = (int(s) for s in out.shape)
    C = int(v.shape[1])
    lib = _lib.load()
    fn = lib.sgv3d_broadcast_channels_bf16 if out.dtype == torch.bfloat16 else lib.sgv3d_broadcast_channels
    with torch.cuda.device(out.device), prof("broadcast_channels"):
        rc = fn(B, H * W, C, ld, int(y_coff), v.data_ptr(), out.data_ptr(), _st(out))
    _lib.check(rc, "sgv3d_broadcast_channels")
    return out


def scale_channels(x, gate, out=None):
    """x NHWC [B,H,W,C] * gate [B,C] (SELayer gating)."""
    B, H, W, C = (int(s) for s in x.shape)
    assert tuple(gate.shape) == (B, C) and x.is_contiguous() and gate.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    fn = lib.sgv3d_scale_channels_bf16 if x.dtype == torch.bfloat16 else lib.sgv3d_scale_channels
    with torch.cuda.device(x.device), prof("scale_channels"):
        rc = fn(B, H * W, C, x.data_ptr(), gate.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_scale_channels")
    return out


def copy_channels(x, out, coff=0):
    """out[B,H,W,C] (contiguous) = x[B,H,W,coff:coff+C]."""
    B, H, W, ld = (int(s) for s in x.shape)
    C = int(out.shape[-1])
    assert out.is_contiguous() and out.numel() == B * H * W * C
    with torch.cuda.device(x.device), prof("copy_channels"):
        rc = _lib.load().sgv3d_copy_channels(B, H * W, C, ld, int(coff), x.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_copy_channels")
    return out


def upsample_bilinear2x(x, out=None):
    B, H, W, C = (int(s) for s in x.shape)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(B, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    lib = _lib.load()
    fn = lib.sgv3d_upsample_bilinear2x_bf16 if x.dtype == torch.bfloat16 else lib.sgv3d_upsample_bilinear2x
    with torch.cuda.device(x.device), prof("upsample_bilinear2x"):
        rc = fn(B, H, W, C, x.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_upsample_bilinear2x")
    return out


def add_mul_sigmoid(a, b, c, out=None):
    """a + b * sigmoid(c), same-shape contiguous tensors."""
    assert a.shape == b.shape == c.shape and a.is_contiguous() and b.is_contiguous() and c.is_contiguous()
    assert a.dtype == b.dtype == c.dtype
    if out is None:
        out = torch.empty_like(a)
    lib = _lib.load()
    fn = lib.sgv3d_add_mul_sigmoid_bf16 if a.dtype == torch.bfloat16 else lib.sgv3d_add_mul_sigmoid
    with torch.cuda.device(a.device), prof("add_mul_sigmoid"):
        rc = fn(a.numel(), a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), _st(a))
    _lib.check(rc, "sgv3d_add_mul_sigmoid")
    return out


def bsm_compose(height_context, semantic_logits, D, ctx, sem, thr):
    """In place on height_context [B,H,W,ld]; semantic_logits [B,H,W,>=sem]."""
    B, H, W, ld = (int(s) for s in height_context.shape)
    with torch.cuda.device(height_context.device), prof("bsm_compose"):
        rc = _lib.load().sgv3d_bsm_compose(B, H * W, int(D), int(ctx), int(sem), ld, semantic_logits.data_ptr(),
                                          int(semantic_logits.shape[-1]), float(thr), height_context.data_ptr(),
                                          _st(height_context))
    _lib.check(rc, "sgv3d_bsm_compose")
    return height_context


DCN_FUSED = _os.environ.get("SGV3D_DCN_FUSED", "1") != "0"     # 0: deformable im2col + one GEMM per group (rounds 1-4)


def deform_conv3x3_eligible(x, convs):
    """f32 NHWC input, packed group weights that share one geometry, channels per group a multiple of 32: what
    sgv3d_deform_conv3x3_forward covers (bf16-activation mode keeps the im2col form)."""
    cpg9 = convs[0].cin
    return (DCN_FUSED and not MFMA_BF16 and not MFMA_F32X3 and x.dtype == torch.float32 and len(convs) <= 8 and cpg9 % 9 == 0
            and (cpg9 // 9) % 32 == 0 and convs[0].cout % 4 == 0 and convs[0].k_order == 1
            and all(c.cin == cpg9 and c.cout == convs[0].cout and c.k_pad == convs[0].k_pad and c.cout_pad == convs[0].cout_pad
                    and c.scale is None and c.shift is None and not c.relu for c in convs))


def deform_conv3x3(x, offset, convs, out=None, y_coff=0):
    """DCNv1 forward (lss_fpn.py:190-198) in one launch: x f32 NHWC [B,H,W,C], offset f32 [B,H,W,>=18], ``convs`` the per-group
    PackedConvs of the [opg, 9 * cpg] matrices (k = tap * cpg + ci).  Writes out[..., y_coff : y_coff + groups * opg]."""
    import ctypes
    B, H, W, C = (int(v) for v in x.shape)
    g, opg = len(convs), convs[0].cout
    assert x.is_contiguous() and offset.is_contiguous() and offset.dtype == torch.float32 and int(offset.shape[-1]) >= 18
    if out is None:
        out = torch.empty(B, H, W, g * opg, dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape[:3]) == (B, H, W)
    ptrs = (ctypes.c_void_p * g)(*[c.w.data_ptr() for c in convs])
    flops = 2.0 * B * H * W * g * opg * (convs[0].cin_real)
    nbytes = 4.0 * (B * H * W * (C + 18 + g * opg) + g * opg * convs[0].cin_real)
    with torch.cuda.device(x.device), prof("conv_dcn_fused", flops, nbytes,
                                           {"symbol": "dcn3x3_fused_kernel", "mfma_flops": 2.0 * B * H * W * g * convs[0].cout * convs[0].cin}
                                           if PROFILE is not None else None):
        rc = _lib.load().sgv3d_deform_conv3x3_forward(B, H, W, C, g, opg, x.data_ptr(), offset.data_ptr(), int(offset.shape[-1]), ptrs,
                                                     convs[0].k_pad, convs[0].cout_pad, out.data_ptr(), int(out.shape[-1]), int(y_coff),
                                                     _st(x))
    _lib.check(rc, "sgv3d_deform_conv3x3_forward")
    return out


def deform_im2col3x3(x, offset, groups, out=None):
    """x NHWC [B,H,W,C], offset NHWC [B,H,W,>=18] -> col [B,H,W,groups*9*(C/groups)]."""
    B, H, W, C = (int(s) for s in x.shape)
    if out is None:
        out = torch.empty(B, H, W, 9 * C, dtype=x.dtype, device=x.device)
    assert offset.dtype == torch.float32 and out.dtype == x.dtype
    lib = _lib.load()
    fn = lib.sgv3d_deform_im2col3x3_bf16 if x.dtype == torch.bfloat16 else lib.sgv3d_deform_im2col3x3
    with torch.cuda.device(x.device), prof("deform_im2col3x3"):
        rc = fn(B, H, W, C, int(groups), x.data_ptr(), offset.data_ptr(), int(offset.shape[-1]), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_deform_im2col3x3")
    return out


def head_final_conv(hidden, weight, bias, branch_of_out, num_branches, hidden_ch, out=None):
    """hidden [nb, B, H, W, hc] (one NHWC map per branch); weight [sum_c,3,3,hc]; -> NCHW [B,sum_c,H,W]."""
    nb, B, H, W, hc = (int(s) for s in hidden.shape)
    assert nb == num_branches and hc == hidden_ch and hidden.is_contiguous()
    total = int(weight.shape[0])
    if out is None:
        out = torch.empty(B, total, H, W, dtype=torch.float32, device=hidden.device)
    with torch.cuda.device(hidden.device), prof("head_final_conv", 2.0 * B * H * W * total * 9 * hidden_ch):
        rc = _lib.load().sgv3d_head_final_conv(B, H, W, int(num_branches), int(hidden_ch), total,
                                              hidden.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                              branch_of_out.data_ptr(), out.data_ptr(), _st(hidden))
    _lib.check(rc, "sgv3d_head_final_conv")
    return out


def centerhead_branches(x, first, w2, b2, out_begin, num_branches, out=None):
    """Both layers of all CenterHead branches in one kernel (hidden maps never leave the chip).
    x NHWC [B,H,W,ld]; first: PackedConv of the concatenated first layers (needs Winograd weights, hidden 64);
    w2 [sum_c,3,3,64]; b2 [sum_c]; out_begin int32 [nb+1] (device).  -> NCHW [B,sum_c,H,W]."""
    B, H, W, ld = (int(s) for s in x.shape)
    assert x.is_contiguous() and first.wino_ok and first.cout == num_branches * 64
    total = int(w2.shape[0])
    if out is None:
        out = torch.empty(B, total, H, W, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nws = lib.sgv3d_centerhead_branches_workspace_bytes(B, H, W, total)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    flops = 2.0 * B * H * W * (first.cout * first.cin * 9 + total * 9 * 64)
    with torch.cuda.device(x.device), prof("conv_wino_head", flops):
        rc = lib.sgv3d_centerhead_branches_forward(B, H, W, first.cin, ld, 0, x.data_ptr(), int(num_branches),
                                                   first.w_wino.data_ptr(), _lib.ptr(first.scale), _lib.ptr(first.shift),
                                                   total, w2.data_ptr(), b2.data_ptr(), out_begin.data_ptr(), out.data_ptr(),
                                                   ws.data_ptr(), nws, _st(x))
    _lib.check(rc, "sgv3d_centerhead_branches_forward")
    return out


def pack_centerhead_f4(w1):
    """First layers of the CenterHead branches concatenated, f32 OIHW [nb*64, 64, 3, 3] (device) -> the F(4x4) fragment
    stream of sgv3d_centerhead_branches_forward_f4 (csrc/head_wino4.hip)."""
    assert w1.is_cuda and w1.dtype == torch.float32 and tuple(w1.shape[1:]) == (64, 3, 3) and int(w1.shape[0]) % 64 == 0
    nb = int(w1.shape[0]) // 64
    lib = _lib.load()
    w1 = w1.contiguous()
    u = torch.empty(int(lib.sgv3d_centerhead_f4_weight_floats(nb)), dtype=torch.float32, device=w1.device)
    with torch.cuda.device(w1.device):
        _lib.check(lib.sgv3d_centerhead_f4_pack_weight(w1.data_ptr(), nb, u.data_ptr(), _st(w1)), "sgv3d_centerhead_f4_pack_weight")
    return u


def centerhead_branches_f4(x, u, scale1, shift1, w2, b2, out_begin, num_branches, out=None):
    """``centerhead_branches`` with the first layers in Winograd F(4x4) form: x NHWC f32 [B,H,W,ld] (channels [0, 64)),
    ``u`` from ``pack_centerhead_f4``, scale1 / shift1 the folded BN of the first layers [nb*64].  -> NCHW [B,sum_c,H,W]."""
    B, H, W, ld = (int(s) for s in x.shape)
    assert x.is_contiguous() and x.dtype == torch.float32
    total = int(w2.shape[0])
    if out is None:
        out = torch.empty(B, total, H, W, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nws = lib.sgv3d_centerhead_branches_workspace_bytes(B, H, W, total)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    flops = 2.0 * B * H * W * (num_branches * 64 * 64 * 9 + total * 9 * 64)
    with torch.cuda.device(x.device), prof("conv_head_wino4", flops):
        rc = lib.sgv3d_centerhead_branches_forward_f4(B, H, W, 64, ld, 0, x.data_ptr(), int(num_branches), u.data_ptr(),
                                                      _lib.ptr(scale1), _lib.ptr(shift1), total, w2.data_ptr(), b2.data_ptr(),
                                                      out_begin.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, _st(x))
    _lib.check(rc, "sgv3d_centerhead_branches_forward_f4")
    return out


def pack_centerhead_bf16(w1, w2, out_begin):
    """First layers of the CenterHead branches concatenated, f32 [nb*64, 64, 3, 3], and final layers f32 [sum_c, 3, 3, 64]
    with ``out_begin`` int32 [nb+1] -> the two bf16 buffers sgv3d_centerhead_branches_forward_bf16 streams
    (fragment-ordered first layers; [branch][tap][4][64] final layers)."""
    w1 = w1.detach().float().contiguous()
    w2 = w2.detach().float().contiguous()
    nb = int(w1.shape[0]) // 64
    assert int(w1.shape[0]) == nb * 64 and tuple(w1.shape[1:]) == (64, 3, 3), "bf16 fused head: 64 -> 64 3x3 first layers"
    assert tuple(w2.shape[1:]) == (3, 3, 64) and int(out_begin.numel()) == nb + 1 and out_begin.dtype == torch.int32
    lib = _lib.load()
    p1 = torch.empty(lib.sgv3d_centerhead_bf16_weight_bytes(nb), dtype=torch.uint8, device=w1.device)
    p2 = torch.empty(lib.sgv3d_centerhead_bf16_weight2_bytes(nb), dtype=torch.uint8, device=w1.device)
    with torch.cuda.device(w1.device):
        _lib.check(lib.sgv3d_centerhead_bf16_pack_weight(w1.data_ptr(), nb, 64, p1.data_ptr(), _st(w1)),
                   "sgv3d_centerhead_bf16_pack_weight")
        _lib.check(lib.sgv3d_centerhead_bf16_pack_weight2(w2.data_ptr(), out_begin.data_ptr(), nb, p2.data_ptr(), _st(w1)),
                   "sgv3d_centerhead_bf16_pack_weight2")
    p1._keep = (w1, w2)             # the pack kernels read them asynchronously
    return p1, p2


def centerhead_branches_bf16(x, packed, scale1, shift1, b2, out_begin, num_branches, out=None, x_coff=0):
    """bf16-MFMA version of ``centerhead_branches``: x NHWC f32 [B,H,W,ld] (channels [x_coff, x_coff+64)), ``packed`` from
    ``pack_centerhead_bf16``, scale1 / shift1 [nb*64] folded BN, b2 [sum_c] -> NCHW f32 [B,sum_c,H,W]."""
    B, H, W, ld = (int(s) for s in x.shape)
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    total = int(b2.shape[0])
    if out is None:
        out = torch.empty(B, total, H, W, dtype=torch.float32, device=x.device)
    flops = 2.0 * B * H * W * (num_branches * 64 * 64 * 9 + total * 9 * 64)
    lib = _lib.load()
    fwd = lib.sgv3d_centerhead_branches_forward_bf16x if x.dtype == torch.bfloat16 else lib.sgv3d_centerhead_branches_forward_bf16
    with torch.cuda.device(x.device), prof("conv_head_bf16", flops):
        rc = fwd(
            B, H, W, 64, ld, int(x_coff), x.data_ptr(), int(num_branches), packed[0].data_ptr(), scale1.data_ptr(),
            shift1.data_ptr(), total, packed[1].data_ptr(), b2.data_ptr(), out_begin.data_ptr(), out.data_ptr(), _st(x))
    _lib.check(rc, "sgv3d_centerhead_branches_forward_bf16")
    return out


def lift(height_context, D, C, want_prob=False, want_lifted=True, lifted_dtype=None):
    """height_context NHWC [B,fH,fW,D+C] -> (prob [B,D,P] | None, lifted [B,D,P,C] | None).
    ``lifted_dtype=torch.bfloat16`` (bf16 compute mode, C % 4 == 0): the lifted tensor is written as bf16."""
    B, fH, fW, ld = (int(s) for s in height_context.shape)
    assert ld == D + C
    P = fH * fW
    lifted_dtype = lifted_dtype or torch.float32
    prob = torch.empty(B, D, P, dtype=torch.float32, device=height_context.device) if want_prob else None
    lifted = torch.empty(B, D, P, C, dtype=lifted_dtype, device=height_context.device) if want_lifted else None
    lib = _lib.load()
    fn = lib.sgv3d_lift_bf16 if (want_lifted and lifted_dtype == torch.bfloat16) else lib.sgv3d_lift
    with torch.cuda.device(height_context.device), prof("lift"):
        rc = fn(B, P, D, C, height_context.data_ptr(), _lib.ptr(prob), _lib.ptr(lifted), _st(height_context))
    _lib.check(rc, "sgv3d_lift")
    return prob, lifted
