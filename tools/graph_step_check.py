#!/usr/bin/env python3
"""One optimiser step of the small model eagerly and as a GraphedTrainStep replay FROM THE SAME STATE (parameters, AdamW moments, step
counter, BatchNorm buffers restored in between): gradients and updated parameters must agree to the run-to-run noise of the eager
step itself (the deformable-convolution adjoint adds with float atomics).  Prints the worst parameter tensors."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops, synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.train_step import DataParallelAdamW, GraphedTrainStep

if os.environ.get("DTYPE") == "bf16":
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
dev = torch.device("cuda", 0)
bconf, hconf = synthetic.small_conf()
torch.manual_seed(0)
model = BEVHeight(bconf, hconf).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
B = 2
imgs = synthetic.make_images(B, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(B, device=dev, scale=bconf['final_dim'][0] / 864)
boxes, labels = synthetic.make_gt(B, seed=0, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
opt = DataParallelAdamW(model.parameters(), lr=float(os.environ.get("LR", "2e-4")))


def forward_backward():
    loss = model.loss(model.get_targets(boxes, labels), model(imgs, mats))
    loss.backward()
    return loss


def eager():
    opt.zero_grad()
    loss = forward_backward()
    opt.step()
    return float(loss.detach())


def snapshot():
    return ([p.clone() for p, _, _ in opt.flat.buckets], [(m.clone(), v.clone()) for m, v in opt.state], opt.steps,
            {k: v.clone() for k, v in model.state_dict().items() if 'running_' in k or 'num_batches' in k})


def restore(snap):
    ps, st, steps, bufs = snap
    for (p, _, _), q in zip(opt.flat.buckets, ps):
        p.copy_(q)
    for (m, v), (m0, v0) in zip(opt.state, st):
        m.copy_(m0); v.copy_(v0)
    opt.steps = steps
    sd = model.state_dict()
    for k, v in bufs.items():
        sd[k].copy_(v)


for _ in range(3):
    eager()
torch.cuda.synchronize()
snap = snapshot()
runs = {}
for tag in ("eager_a", "eager_b"):
    restore(snap)
    loss = eager()
    torch.cuda.synchronize()
    runs[tag] = (loss, [g.clone() for _, g, _ in opt.flat.buckets], [p.clone() for p, _, _ in opt.flat.buckets])
restore(snap)
graphed = GraphedTrainStep(forward_backward, opt, warmup=0, strict=True)
restore(snap)
loss = float(graphed().detach())
torch.cuda.synchronize()
runs["graph"] = (loss, [g.clone() for _, g, _ in opt.flat.buckets], [p.clone() for p, _, _ in opt.flat.buckets])
restore(snap)
loss = float(graphed().detach())
torch.cuda.synchronize()
runs["graph_again"] = (loss, [g.clone() for _, g, _ in opt.flat.buckets], [p.clone() for p, _, _ in opt.flat.buckets])
print("loss:", {k: v[0] for k, v in runs.items()})
names = {id(p): n for n, p in model.named_parameters()}


def worst(a, b, what):
    out = []
    for bi, (_, _, entries) in enumerate(opt.flat.buckets):
        for p, off, cnt in entries:
            x, y = runs[a][what][bi][off:off + cnt], runs[b][what][bi][off:off + cnt]
            d = float((x - y).abs().max())
            s = float(x.abs().max())
            out.append((d / (s + 1e-30), d, s, names.get(id(p), "?")))
    out.sort(reverse=True)
    return out


for a, b in (("eager_a", "eager_b"), ("eager_a", "graph"), ("graph", "graph_again")):
    for what, label in ((1, "grad"), (2, "param")):
        w = worst(a, b, what)
        print(f"{a} vs {b}, {label}: worst relative {w[0][0]:.3e} ({w[0][3]}), tensors above 1e-4: {sum(1 for r in w if r[0] > 1e-4)} of {len(w)}")
        for r in w[:4]:
            print(f"      {r[0]:.3e}  abs {r[1]:.3e} of {r[2]:.3e}  {r[3]}")

# ... and a run of steps from the same state (the losses of a trajectory amplify any per-step difference)
N = int(os.environ.get("STEPS", "40"))
traces = {}
for tag in ("eager", "eager_again", "graph", "graph_again"):
    restore(snap)
    tr = []
    for _ in range(N):
        tr.append(eager() if tag.startswith("eager") else float(graphed().detach()))
    traces[tag] = tr
for tag, tr in traces.items():
    print(f"{tag:12s}", " ".join(f"{v:.4f}" for v in tr[:3]), "...", " ".join(f"{v:.4f}" for v in tr[-3:]))
rel = lambda a, b: max(abs(x - y) / abs(x) for x, y in zip(traces[a], traces[b]))
print(f"worst relative loss difference over {N} steps: eager vs eager {rel('eager', 'eager_again'):.2e}, eager vs graph {rel('eager', 'graph'):.2e}, "
      f"graph vs graph {rel('graph', 'graph_again'):.2e}")
