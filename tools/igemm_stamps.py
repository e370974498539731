"""Where does a workgroup of the f32 implicit-GEMM kernel spend its cycles on the short-K 1x1 layers of cfg-2?
Builds a private copy of the library with -DSGV3D_IGEMM_STAMPS (every 97th workgroup writes 4 cycle-counter stamps) under
gpurun_out/, loads it instead of the product library, and prints per layer: prologue (entry -> first k-tile in LDS), k loop,
epilogue issue, and how the stamped workgroups' lifetimes spread over the kernel."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "sgv3d_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "stamps_build")
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, "conv_igemm_stamps.o")
lib = os.path.join(out, "libsgv3d_hip_stamps.so")
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DSGV3D_IGEMM_STAMPS", "-c", os.path.join(csrc, "conv_igemm.hip"), "-o", obj])
others = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "conv_igemm.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, obj, *others])
print("built", lib, flush=True)

import torch
from sgv3d_amd import _lib
_lib.LIB_PATH = lib
from sgv3d_amd import hip_ops
L = _lib.load()
L.sgv3d_igemm_debug_stamps.restype = ctypes.c_int
L.sgv3d_igemm_debug_stamps.argtypes = [ctypes.c_void_p]
DEV = "cuda:0"
# (B, H, W, cin, cout, residual): ResNet-50 1x1 layers of cfg-2
SHAPES = [(1, 216, 384, 64, 256, True), (1, 216, 384, 256, 64, False), (1, 108, 192, 128, 512, True),
          (1, 108, 192, 512, 128, False), (1, 54, 96, 256, 1024, True), (1, 54, 96, 1024, 256, False), (1, 108, 192, 512, 256, False)]
for B, H, W, cin, cout, res in SHAPES:
    w = torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5
    conv = hip_ops.PackedConv(w, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV)
    r = torch.randn(B, H, W, cout, device=DEV) if res else None
    y = torch.empty(B, H, W, cout, device=DEV)
    for _ in range(3):
        conv(x, y, residual=r, tile=4, split_k=1)
    dbg = torch.zeros(64 * 4, dtype=torch.int64, device=DEV)
    assert L.sgv3d_igemm_debug_stamps(dbg.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); conv(x, y, residual=r, tile=4, split_k=1); e1.record()
    torch.cuda.synchronize()
    L.sgv3d_igemm_debug_stamps(None)
    t = dbg.cpu().view(64, 4)
    rows = [tuple(int(v) for v in q) for q in t if int(q[0]) > 0 and int(q[3]) > 0]
    wgs = -(-B * H * W // 64) * -(-cout // 64)
    if not rows:
        print(cin, cout, "no stamps"); continue
    t0 = min(q[0] for q in rows)
    pro = sorted(q[1] - q[0] for q in rows); loop = sorted(q[2] - q[1] for q in rows); epi = sorted(q[3] - q[2] for q in rows)
    life = sorted(q[3] - q[0] for q in rows)
    med = lambda v: v[len(v) // 2]
    nk = cin // 32
    print(f"{H}x{W} {cin}->{cout}{' +res' if res else ''}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {wgs} workgroups ({wgs / 1024:.2f} rounds of 1024 slots), "
          f"{len(rows)} stamped; median cycles: prologue {med(pro)}, k loop {med(loop)} ({nk} k-tiles, {nk * 1024} MFMA cycles per wave), "
          f"epilogue {med(epi)}, lifetime {med(life)} (min {life[0]}, max {life[-1]}); starts at {sorted(q[0] - t0 for q in rows)[::max(1, len(rows) // 8)]}", flush=True)
