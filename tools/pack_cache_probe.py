import os, sys, torch
sys.path.insert(0, os.getcwd())
from sgv3d_amd import hip_ops, synthetic, pack_cache
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.train_step import DataParallelAdamW, reference_lr
hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
dev = torch.device("cuda", 0)
bconf, hconf = synthetic.r50_256_conf()
torch.manual_seed(0)
model = BEVHeight(bconf, hconf).to(dev).train()
B = 2
imgs = synthetic.make_images(B, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(B, device=dev)
boxes, labels = synthetic.make_gt(B, seed=0, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
opt = DataParallelAdamW(model.parameters(), lr=reference_lr(B, 1), max_grad_norm=5.0)
for _ in range(3):
    opt.zero_grad()
    loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats)); loss.backward(); opt.step()
torch.cuda.synchronize()
c = opt.packs
jobs = [(e, n, j) for e in c.entries.values() if e.tracked for n, j in e.jobs.items()]
tot = sum(j.dst.numel() for _, _, j in jobs)
print("entries", len(c.entries), "tracked", sum(e.tracked for e in c.entries.values()), "jobs", len(jobs), "elements", tot / 1e6, "M; bf16 jobs", sum(j.bf16 for _, _, j in jobs))
untr = [(tuple(e.param.shape)) for e in c.entries.values() if not e.tracked]
print("untracked", len(untr), untr[:10])
import collections
print(collections.Counter(n for _, n, _ in jobs))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    c.refresh(dev); ev[i + 1].record()
torch.cuda.synchronize()
print("refresh us", sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(10))[5])
big = sorted(jobs, key=lambda t: -t[2].dst.numel())[:8]
for e, n, j in big:
    print(n, tuple(e.param.shape), j.dst.numel())
