#!/usr/bin/env python3
"""Out-of-bounds write hunt: every torch.empty / new_empty made during one eager forward gets a 64 KiB sentinel
tail; after the forward the tails are checked and offenders reported with their allocation site."""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight

cfg = os.environ.get("CFG", "cfg2")
bc, hc = {"cfg2": S.r50_256_conf, "cfg5": S.bsm_r101_256_conf, "small": S.small_conf}[cfg]()
torch.manual_seed(0)
m = BEVHeight(bc, hc).eval(); S.randomize_norm_stats_(m, 0); m = m.cuda()
B = int(os.environ.get("BATCH", "1"))
img = S.make_images(B, bc['final_dim'], device='cuda', seed=1)
mats = S.make_mats(B, device='cuda')
with torch.no_grad():
    m(img, mats)                    # packs weights, autotunes
torch.cuda.synchronize()

GUARD = 16384                       # elements
records = []
_empty = torch.empty

def guarded_empty(*size, **kw):
    dev = kw.get("device", None)
    if dev is None or not str(dev).startswith("cuda"):
        return _empty(*size, **kw)
    shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
    n = 1
    for v in shape:
        n *= int(v)
    flat = _empty(n + GUARD, **kw)
    tail = flat[n:]
    if flat.dtype.is_floating_point:
        tail.fill_(-12345.0)
    else:
        tail.fill_(77)
    site = "".join(traceback.format_stack(limit=6)[:-1])
    records.append((flat, n, site))
    return flat[:n].view(*shape)

torch.empty = guarded_empty
_new_empty = torch.Tensor.new_empty
def guarded_new_empty(self, *size, **kw):
    kw.setdefault("dtype", self.dtype); kw.setdefault("device", self.device)
    return guarded_empty(*size, **kw)
torch.Tensor.new_empty = guarded_new_empty
with torch.no_grad():
    for _ in range(2):
        m(img, mats)
torch.cuda.synchronize()
torch.empty = _empty
torch.Tensor.new_empty = _new_empty
bad = 0
for flat, n, site in records:
    tail = flat[n:]
    ok = bool((tail == (-12345.0 if flat.dtype.is_floating_point else 77)).all())
    if not ok:
        bad += 1
        nz = (tail != (-12345.0 if flat.dtype.is_floating_point else 77)).nonzero().flatten()
        print(f"OVERFLOW past a {n}-element {flat.dtype} buffer: {nz.numel()} guard elements clobbered, first at +{int(nz[0])}, last at +{int(nz[-1])}\n{site}")
print(f"guarded allocations: {len(records)}, overflows: {bad}")
sys.exit(1 if bad else 0)
