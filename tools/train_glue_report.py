#!/usr/bin/env python3
"""Where the torch "glue" launches of a cfg-2 training step come from (copies, adds, fills, cats between the HIP kernels): one eager
step under torch.profiler with Python stacks, the aten operators that launch a kernel or a device copy grouped by the sgv3d_amd source
line that called them.  DTYPE=bf16 for the mixed-precision step."""
import collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops, synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.train_step import DataParallelAdamW, reference_lr

if os.environ.get("DTYPE", "f32") == "bf16":
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
dev = torch.device("cuda", 0)
bconf, hconf = synthetic.r50_256_conf()
torch.manual_seed(0)
model = BEVHeight(bconf, hconf).to(dev).train()
B = int(os.environ.get("BATCH", "2"))
imgs = synthetic.make_images(B, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(B, device=dev)
boxes, labels = synthetic.make_gt(B, seed=0, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
opt = DataParallelAdamW(model.parameters(), lr=reference_lr(B, 1))


def step():
    opt.zero_grad()
    loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats))
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

# Python-level sources of copies: .contiguous() on a non-contiguous tensor, clone, copy_, pad, cat, sum (C++-internal ones -- autograd's own
# accumulations and view gradients -- do not pass here)
import traceback
SITES = collections.Counter()
root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    for fr in reversed(traceback.extract_stack(limit=12)[:-2]):
        if "sgv3d_amd/" in fr.filename and "train_glue_report" not in fr.filename:
            return f"{fr.filename.replace(root_dir + '/', '')}:{fr.lineno}"
    return "?"


def wrap(owner, name, cond=lambda *a, **k: True):
    orig = getattr(owner, name)

    def f(*a, **k):
        if cond(*a, **k):
            SITES[(f"{getattr(owner, '__name__', owner)}.{name}", site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


wrap(torch.Tensor, "contiguous", lambda self, *a, **k: self.is_cuda and not self.is_contiguous())
wrap(torch.Tensor, "clone", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "copy_", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "sum", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "__setitem__", lambda self, *a, **k: self.is_cuda)
wrap(torch.nn.functional, "pad")
wrap(torch, "cat")
wrap(torch, "zeros")
step()
torch.cuda.synchronize()
print("--- Python-level copy / reduction sources in one step")
for (op, where), n in SITES.most_common(40):
    print(f"  n={n:4d}  {op:28s} {where}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev_time(e):
    for a in ("self_device_time_total", "self_cuda_time_total"):
        v = getattr(e, a, None)
        if v:
            return float(v)
    return 0.0


agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_stack_n=12):
    t = dev_time(e)
    if t <= 0 or not e.key.startswith("aten::"):
        continue
    site = next((s for s in (e.stack or []) if "sgv3d_amd/" in s or "tools/" in s), (e.stack or ["?"])[0])
    k = (e.key, site.replace(root + "/", "").strip()[:120])
    agg[k][0] += e.count
    agg[k][1] += t
tot = sum(v[1] for v in agg.values())
print(f"aten operators with device time of their own in one step: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.2f} ms")
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "40"))]:
    print(f"{us / 1e3:7.3f} ms  n={n:4d}  {name:26s} {site}")
