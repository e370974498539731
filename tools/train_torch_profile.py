import os, sys, torch
sys.path.insert(0, os.getcwd())
from sgv3d_amd import hip_ops, synthetic
from sgv3d_amd.models.bev_height import BEVHeight
dev = torch.device("cuda", 0)
bconf, hconf = synthetic.r50_256_conf()
torch.manual_seed(0)
model = BEVHeight(bconf, hconf).to(dev).train()
imgs = synthetic.make_images(2, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(2, device=dev)
boxes, labels = synthetic.make_gt(2, seed=0, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
def step():
    for p in model.parameters(): p.grad = None
    preds = model(imgs, mats)
    loss = model.loss(model.get_targets(boxes, labels), preds)
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(((getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0), e.key, e.count) for e in ka), reverse=True)
tot = 0
for t, k, n in rows[:45]:
    print(f"{t/1e3:8.2f} ms n={n:4d} {k[:110]}")
