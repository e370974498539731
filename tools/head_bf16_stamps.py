"""Where does a branch iteration of head_bf16_kernel spend its cycles?  (cycle-counter stamps of workgroup 0)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import _lib, hip_ops
lib = _lib.load()
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
counts = []
for nc in (1, 2, 2, 1, 2, 2):
    counts += [2, 1, 3, 2, 2, nc]
nb = len(counts)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, H, W, 64, generator=g).cuda()
w1 = (torch.randn(nb * 64, 64, 3, 3, generator=g) / 24).cuda()
w2 = (torch.randn(sum(counts), 3, 3, 64, generator=g) / 24).cuda()
ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
packed = hip_ops.pack_centerhead_bf16(w1, w2, ob)
sc, sh, b2 = torch.ones(nb * 64).cuda(), torch.zeros(nb * 64).cuda(), torch.zeros(sum(counts)).cuda()
def timed(tag):
    for _ in range(3):
        hip_ops.centerhead_branches_bf16(x, packed, sc, sh, b2, ob, nb)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    evs[0].record()
    for i in range(5):
        hip_ops.centerhead_branches_bf16(x, packed, sc, sh, b2, ob, nb)
        evs[i + 1].record()
    torch.cuda.synchronize()
    print(f"{tag}: {min(evs[i].elapsed_time(evs[i + 1]) for i in range(5)) * 1e3:.0f} us")
timed("warp-specialised kernel (default)")
lib.sgv3d_centerhead_bf16_select_plain(1)
timed("single-role kernel")
lib.sgv3d_centerhead_bf16_select_plain(0)
dbg = torch.zeros(4 * nb + 2, dtype=torch.int64).cuda()
lib.sgv3d_centerhead_bf16_debug_stamps(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip_ops.centerhead_branches_bf16(x, packed, sc, sh, b2, ob, nb); e1.record()
torch.cuda.synchronize()
lib.sgv3d_centerhead_bf16_debug_stamps(None)
t = dbg.cpu().tolist()
print(f"kernel {e0.elapsed_time(e1)*1e3:.0f} us; workgroup 0: {t[4*nb]-t[0]} counter ticks for {nb} branches")
for br in (0, 1, 17, 35):
    s = t[1 + 4 * br: 5 + 4 * br]
    prev = t[0] if br == 0 else t[4 * br]
    print(f"branch {br}: layer1 {s[0]-prev}  wait-barrier {s[1]-s[0]}  epilogue+barrier {s[2]-s[1]}  layer2+store {s[3]-s[2]}")
tot = [0, 0, 0, 0]
for br in range(nb):
    s = t[1 + 4 * br: 5 + 4 * br]; prev = t[0] if br == 0 else t[4 * br]
    for i, v in enumerate((s[0]-prev, s[1]-s[0], s[2]-s[1], s[3]-s[2])): tot[i] += v
print("sum over branches:", dict(zip(("layer1", "barrier", "epilogue", "layer2"), tot)))
