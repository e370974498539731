#!/usr/bin/env python3
"""Training step of BASELINE config 4's per-GPU share on synthetic data: forward (train mode) + targets + loss +
backward + gradient all-reduce + fused AdamW.  `python tools/train_bench.py [--batch 4] [--steps 5] [--config cfg2|cfg5]`
or, BASELINE config 4 (global batch 32 = 4 per GPU over 8 MI355X, 304 MB of fp32 gradients all-reduced over RCCL/xGMI in
flat 48 MiB buckets launched from inside backward, clipped to a global gradient norm of 5 as the reference's Trainer does):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/train_bench.py --config cfg4 --steps 10  Prints one JSON line (samples/s over all ranks); this is
NOT the headline metric of bench.py (inference frames/s), it tracks SURVEY §8(f) rank 2."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops, synthetic
from sgv3d_amd.models.bev_height import BEVHeight
from sgv3d_amd.replicas import ReplicaGroup
from sgv3d_amd.train_step import DataParallelAdamW, reference_lr

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4", "cfg5", "small"],
                help="cfg4 = the cfg-2 model at BASELINE configs[3]'s per-GPU share (batch 4 per rank, global batch 4 x ranks)")
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                help="bf16 = mixed-precision step: every convolution product (forward, data gradient, weight gradient) on the bf16 matrix "
                     "cores with f32 accumulation; parameters, gradients, activations, BatchNorm, loss and AdamW stay f32")
ap.add_argument("--no-overlap", action="store_true", help="all-reduce after backward instead of from grad hooks")
ap.add_argument("--no-dropout", action="store_true", help="Dropout(p=0) instead of the training default 0.5 (comparisons that must not depend on the random stream)")
ap.add_argument("--trace-loss", action="store_true", help="record the loss of every timed step (a host read per step: not for timing)")
ap.add_argument("--graph", action="store_true", help="the whole step (forward, targets, loss, backward, AdamW) as ONE hipGraph replay "
                "(train_step.GraphedTrainStep); single process only: with a process group the gradient all-reduce stays outside the graph")
ap.add_argument("--clip", type=float, default=5.0, help="global gradient-norm bound (Lightning's gradient_clip_val of the reference's Trainer, "
                "exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:405); 0 = off")
ap.add_argument("--bucket-mib", type=int, default=None, help="flat gradient bucket size (default train_step.DEFAULT_BUCKET_BYTES = 48 MiB)")
ap.add_argument("--profile", action="store_true", help="per-kernel-family times of one step (HIP events, eager)")
args = ap.parse_args()

if args.dtype == "bf16":
    hip_ops.MFMA_BF16 = True
    hip_ops.BF16_ACTIVATIONS = False        # f32 tensors between the layers (the training kernels' contract); bf16 operands only
local = int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
group = ReplicaGroup(device=dev)
if args.config == "cfg4":
    args.batch = 4
bconf, hconf = {"cfg2": synthetic.r50_256_conf, "cfg4": synthetic.r50_256_conf, "cfg5": synthetic.bsm_r101_256_conf, "small": synthetic.small_conf}[args.config]()
BSM = bool(bconf.get('is_bsm'))                 # cfg5: SGV3D BSM R101 with the semantic (SAM-mask) supervision
if BSM:
    bconf = dict(bconf, is_train_height=True)
torch.manual_seed(0)
model = BEVHeight(bconf, hconf, is_train_height=BSM).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0 if args.no_dropout else 0.5
if args.config == "small":
    model.head.train_cfg = dict(model.head.train_cfg, grid_size=[256, 256, 1], point_cloud_range=[0, -12.8, -5, 25.6, 12.8, 3])
imgs = synthetic.make_images(args.batch, final=bconf['final_dim'], device=dev, seed=group.rank)
mats = synthetic.make_mats(args.batch, device=dev, scale=bconf['final_dim'][0] / 864)      # (calibration of the image size in use)
boxes, labels = synthetic.make_gt(args.batch, seed=group.rank, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]
if BSM:
    from sgv3d_amd.losses import SemanticSupervision
    semantic = SemanticSupervision(8)
    gt_semantic = torch.randint(0, 7, (args.batch, 1) + tuple(bconf['final_dim']), dtype=torch.uint8,
                                generator=torch.Generator().manual_seed(group.rank)).to(dev)
opt = DataParallelAdamW(model.parameters(), lr=reference_lr(args.batch, group.world), max_grad_norm=args.clip or None,
                        bucket_bytes=None if args.bucket_mib is None else args.bucket_mib << 20)   # broadcasts rank 0's parameters
if not args.no_overlap:
    opt.overlap_with_backward()
nparam = sum(p.numel() for p in model.parameters())


def forward_backward():
    preds = model(imgs, mats)
    if BSM:                                     # exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:304-330
        preds, img_preds = preds
    targets = model.get_targets(boxes, labels)
    loss = model.loss(targets, preds)
    if BSM:
        loss = loss + semantic(img_preds, gt_semantic) * 500
    if BW_EVENTS:
        BW_EVENTS[0].record()
    loss.backward()
    if BW_EVENTS:
        BW_EVENTS[1].record()
    EARLY[0] = len(getattr(opt, '_early', {}))
    return loss


def step():
    opt.zero_grad()
    loss = forward_backward()
    opt.all_reduce_grads()
    opt.step()
    return loss


EARLY = [0]
BW_EVENTS = []


loss = None
for _ in range(args.warmup):
    loss = step().detach()              # (no reference to the autograd graph is kept: GraphedTrainStep)
torch.cuda.synchronize()
LAST = [loss]


def run():                                  # (the reported loss is the LAST timed step's, eagerly and graphed)
    LAST[0] = step().detach()
    return LAST[0]


if args.graph:
    from sgv3d_amd.train_step import GraphedTrainStep
    graphed = GraphedTrainStep(forward_backward, opt, strict=True)
    run = graphed
    loss = run()
    torch.cuda.synchronize()
TRACE = []
if args.trace_loss:
    _run = run

    def run():
        v = _run()
        TRACE.append(float(v.detach()))
        return v
elapsed = group.timed(run, args.steps)
loss = graphed.result if args.graph else LAST[0]
grad_norm, clip_coef = opt.grad_norm(), opt.clip_coefficient()
FIRST = None
if opt._collectives() and not args.no_overlap and getattr(opt, '_hooks', None):
    # one more eager step with three events: start of backward, launch of bucket 0's all-reduce, end of backward
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    opt.first_early_event = ev[1]
    BW_EVENTS[:] = [ev[0], ev[2]]
    try:
        step()
    finally:
        BW_EVENTS[:] = []
        opt.first_early_event = None
    torch.cuda.synchronize()
    if EARLY[0] > 0:
        FIRST = ev[0].elapsed_time(ev[1]) / max(ev[0].elapsed_time(ev[2]), 1e-6)
out = {"metric": "training samples/s (forward + loss + backward + all-reduce + AdamW)", "value": group.world * args.batch * args.steps / elapsed,
       "unit": "samples/s", "n_gpus": group.world, "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps,
       "batch_per_gpu": args.batch, "global_batch": args.batch * group.world, "dtype": args.dtype,
       "world_size": group.dist.get_world_size() if group.dist is not None else 1, "backend": group.backend,
       "allreduce_bytes_per_step": 4 * sum(g.numel() for _, g, _ in opt.flat.buckets), "allreduce_buckets": len(opt.flat.buckets),
       "allreduce_overlapped_with_backward": not args.no_overlap, "parameters": nparam, "loss": float(loss.detach()), "config": args.config, "loss_trace": TRACE or None, "graph": bool(args.graph), "graph_replays": graphed.replays if args.graph else 0,
       "update_in_graph": bool(graphed.in_graph_update) if args.graph else None, "optimizer_steps": opt.steps,
       "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 2**30, "data": "synthetic",
       # (a 1-rank group with SGV3D_FORCE_DIST=1 still broadcasts / all-reduces through RCCL: the single-GPU stand-in for cfg-4)
       "collectives_active": bool(opt._collectives()), "allreduces_launched_inside_backward": EARLY[0],
       "first_allreduce_launch_at_fraction_of_backward": FIRST, "bucket_mib": max(g.numel() for _, g, _ in opt.flat.buckets) * 4 / 2**20,
       "gradient_clip_val": args.clip or None, "grad_norm_last_step": grad_norm, "clip_coefficient_last_step": clip_coef,
       "frozen_parameters": sorted(n for n, p in model.named_parameters() if not p.requires_grad),
       # packed weight forms kept across steps and refreshed by one gather launch per step (sgv3d_amd/pack_cache.py; SGV3D_PACK_CACHE=0: off)
       "pack_cache": (None if getattr(opt, "packs", None) is None else
                      {"forms": len(opt.packs.jobs()), "elements": sum(j.dst.numel() for j in opt.packs.jobs()),
                       "layers_packing_per_call": sum(not e.tracked for e in opt.packs.entries.values())}),
       "param_checksum": float(sum(p.double().abs().sum() for p, _, _ in opt.flat.buckets))}
if args.profile:
    # the profiled step holds collectives (loss-factor and gradient all-reduces): every rank runs it, rank 0 reports
    hip_ops.PROFILE = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    fam = {}
    for name, flops, e0, e1, *_ in hip_ops.PROFILE:
        d = fam.setdefault(name.split('|')[0], [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1)
        d[1] += flops
        d[2] += 1
    hip_ops.PROFILE = None
    out["profiled_step_ms"] = 1e3 * wall
    out["hip_kernels_ms"] = {k: {"ms": round(v[0], 2), "launches": v[2], "tflops": round(v[1] / v[0] / 1e9, 1) if v[0] > 0 and v[1] > 0 else None}
                             for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
hip_ops.save_tune_db()          # no-op unless SGV3D_TUNE_CACHE is set (tools/profile_train.sh)
group.close()
if group.rank == 0:
    print(json.dumps(out))
