#!/usr/bin/env python3
"""Long-run stability soak: N frames through a 3-slot FramePipeline with changing inputs; every 3rd frame is a
repeat of a reference frame and must reproduce its outputs bit for bit.  usage: soak_pipeline.py [frames=600]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.pipeline import FramePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bc, hc = S.r50_256_conf()
torch.manual_seed(0)
m = BEVHeight(bc, hc).eval()
S.randomize_norm_stats_(m, 0)
m = m.cuda()
ref_img = S.make_images(1, bc['final_dim'], device='cuda', seed=1)
other = [S.make_images(1, bc['final_dim'], device='cuda', seed=s) for s in (2, 3)]
mats = S.make_mats(1, device='cuda')
slots = int(os.environ.get('SOAK_SLOTS', '3'))
pipe = FramePipeline(m, ref_img, mats, slots=slots, use_graph=not os.environ.get('SOAK_EAGER'))
slot = pipe.submit(ref_img, mats)
ref = [t.clone() for task in pipe.result(slot) for t in task[0].values()]
bad = 0
t0 = time.time()
pending = []
for i in range(n):
    is_ref = i % 3 == 0
    slot = pipe.submit(ref_img if is_ref else other[i % 2], mats)
    pending.append((slot, is_ref))
    if i % 100 == 0 or os.environ.get('SOAK_VERBOSE'):
        torch.cuda.synchronize()
        print('frame', i, flush=True)
    if len(pending) == slots:
        s0, r0 = pending.pop(0)
        out = pipe.result(s0)
        if r0:
            got = [t for task in out for t in task[0].values()]
            if not all(torch.equal(a, b) for a, b in zip(got, ref)):
                bad += 1
                print("frame mismatch at", i - 2)
        elif not all(torch.isfinite(t).all() for task in out for t in task[0].values()):
            bad += 1
            print("non-finite at", i - 2)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"soak: {n} frames in {dt:.1f} s ({n / dt:.1f} frames/s incl. host checks), failures: {bad}")
sys.exit(1 if bad else 0)
