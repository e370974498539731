#!/usr/bin/env python3
"""Debug: error maps of the F(4x4) fused head for structured weights (which tap / which channel group is wrong)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sgv3d_amd import hip_ops

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

def show(tag, out, ref):
    err = (out - ref).abs()
    print(f"--- {tag}: max err {float(err.max()):.3e} (scale {float(ref.abs().max()):.2f})")
    e = err[0, 0]
    for y in range(e.shape[0]):
        print("".join("." if v < 1e-3 else "X" for v in e[y].tolist()))

g = torch.Generator().manual_seed(1)
H = W = 16
x = torch.randn(1, H, W, 64, generator=g)
# 1. hidden = identity-ish: w1 centre tap delta (hidden[ch] = x[ch]), no relu effect (shift large), final = centre tap of channel c
for ch in (0, 5, 17, 63):
    w1 = torch.zeros(64, 64, 3, 3); 
    for c in range(64): w1[c, c, 1, 1] = 1.0
    sc = torch.ones(64); sh = torch.full((64,), 10.0)
    w2 = torch.zeros(1, 64, 3, 3); w2[0, ch, 1, 1] = 1.0
    out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
    show(f"identity hidden, final = centre tap of channel {ch}", out, ref)
# 2. all channels summed at centre tap
w2 = torch.zeros(1, 64, 3, 3); w2[0, :, 1, 1] = 1.0
out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
show("identity hidden, final = sum of all channels (centre tap)", out, ref)
# 3. single channel, each tap
for ky in range(3):
    for kx in range(3):
        w2 = torch.zeros(1, 64, 3, 3); w2[0, 9, ky, kx] = 1.0
        out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
        show(f"identity hidden, final = tap ({ky},{kx}) of channel 9", out, ref)
# 4. random first layer, centre tap final of one channel
w1 = torch.randn(64, 64, 3, 3, generator=g) / 24.0
w2 = torch.zeros(1, 64, 3, 3); w2[0, 9, 1, 1] = 1.0
out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
show("random first layer, final = centre tap of channel 9", out, ref)
# 5. 32x32 (4 blocks): ring
H = W = 32
x = torch.randn(1, H, W, 64, generator=g)
w2 = torch.randn(1, 64, 3, 3, generator=g) / 24.0
out, ref, hid = run(H, W, w1, sc, sh, w2, torch.zeros(1), (1,), x)
show("random everything 32x32", out, ref)
