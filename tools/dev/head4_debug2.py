import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
torch.set_printoptions(precision=3, linewidth=250, sci_mode=False)
import importlib
from sgv3d_amd import hip_ops
import torch.nn.functional as F

def run(H, W, w1, sc, sh, w2, b2, counts, x):
    nb = len(counts)
    hid = (F.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), padding=1) * sc.double()[None, :, None, None]
           + sh.double()[None, :, None, None]).clamp_min(0)
    ref, off = [], 0
    for k, c in enumerate(counts):
        ref.append(F.conv2d(hid[:, k * 64:(k + 1) * 64], w2[off:off + c].double(), b2[off:off + c].double(), padding=1))
        off += c
    ref = torch.cat(ref, 1)
    ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).cuda()
    out = hip_ops.centerhead_branches_f4(x.cuda(), hip_ops.pack_centerhead_f4(w1.cuda()), sc.cuda(), sh.cuda(),
                                         w2.permute(0, 2, 3, 1).contiguous().cuda(), b2.cuda(), ob, nb)
    return out.cpu().double(), ref, hid

H = W = 16
# x[y, x, ch] = 100*ch + 10*y... make it decodable: value = ch + y/16 + x/256
ys, xs, cs = torch.meshgrid(torch.arange(16.), torch.arange(16.), torch.arange(64.), indexing="ij")
x = (cs + ys / 16 + xs / 256)[None]
w1 = torch.zeros(64, 64, 3, 3)
for c in range(64): w1[c, c, 1, 1] = 1.0
sc = torch.ones(64); sh = torch.zeros(64)
for ch in (0, 1, 4, 5, 16, 17):
    w2 = torch.zeros(1, 64, 3, 3); w2[0, ch, 1, 1] = 1.0
    out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
    print("channel", ch, "expected ch + y/16 + x/256; got (rows 0..5, cols 0..7):")
    print(out[0, 0, :6, :8])
