#!/usr/bin/env python3
"""What one calibration refresh launches (sgv3d_amd/calibration.py device path: new tensor objects, same numbers), cfg-2 model:
kernel names and counts per call from torch.profiler, and the refresh's time on an otherwise idle device."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.pipeline import eager_forward
from torch.profiler import profile, ProfilerActivity
from torch.autograd import DeviceType

dev = torch.device("cuda", 0)
bconf, hconf = synthetic.r50_256_conf()
torch.manual_seed(0)
m = BEVHeight(bconf, hconf).to(dev).eval()
imgs = synthetic.make_images(1, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(1, device=dev)
with torch.no_grad(), eager_forward(m):
    m(imgs, mats)
    fresh = [{k: v.clone() for k, v in mats.items()} for _ in range(12)]
    for f in fresh[:2]:
        m.backbone.calibration(f, 0)
    torch.cuda.synchronize()
    N = 10
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for f in fresh[2:2 + N]:
            m.backbone.calibration(f, 0)
        torch.cuda.synchronize()
    c = collections.Counter()
    t = collections.Counter()
    for e in prof.events():
        if e.device_type == DeviceType.CUDA:
            c[e.name[:90]] += 1
            t[e.name[:90]] += e.device_time
    print(f"per refresh: {sum(c.values()) / N:.1f} device operations, {sum(t.values()) / N:.1f} us of device time")
    for k, n in c.most_common():
        print(f"  {n / N:5.1f} x {t[k] / n:6.1f} us  {k}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for f in fresh[2:2 + N]:
        m.backbone.calibration(f, 0)
    e1.record()
    torch.cuda.synchronize()
    print(f"wall per refresh (eager launches from Python, idle device): {e0.elapsed_time(e1) / N * 1e3:.0f} us")
