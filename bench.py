#!/usr/bin/env python3
"""Headline benchmark: camera frames/s of the BEVHeight camera->BEV forward on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg-2"): ResNet-50, 864x1536 image -> 256x256 BEV, batch 1 per
GPU, fp32, synthetic images / DAIR-like calibration / random-init weights (no dataset or checkpoint
is reachable), inputs resident in HBM before the timed region.  One "step" = one full forward
(image backbone + neck + HeightNet + lift + geometry + voxel pooling + BEV head) of the per-GPU batch;
frames shard over GPUs as independent replicas (no collective on the data path), so
``value`` = N * batch * K / max-over-ranks(time) and ``scaling`` is weak.

Printed by rank 0 as ONE JSON line, with
* ``value``: the K timed steps with the calibration handed over as FRESH tensor objects every frame -- the reference harness's
  call pattern (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:244-247); ``cached_calibration_value`` beside it
  (a caller that keeps its calibration tensors; rounds 1-5 reported that one), ``native_f32_value`` (a child run with every
  product on the f32 MFMA; ``config.products`` says which products of ``value`` are assembled from bf16 partial products);
* ``roofline``: the dominant MFMA kernel SYMBOL of an instrumented pass over the same K steps (one stream, eager launches) --
  executed flops of its launches / their durations from roctracer's kernel timestamps (torch.profiler), the quantity a rocprofv3
  --kernel-trace reports; the HIP-event measurement of the same launches (raw and with the calibrated marker gap subtracted)
  beside it with the ratio of the two; ``frac_from_rocprof``: the same flops per launch over the symbol's average in the
  committed trace of ``bench.py --roofline-only`` (profiles/r*_bench_roofline_kernel_stats.csv), refused when that trace holds
  another launch mix; ``by_symbol``: every MFMA kernel whose executed flops the pass recorded (the f32x3 GEMMs against the bf16
  peak with the bf16 flops they execute);
* ``roofline_hbm``: the voxel-pooling operator against HBM peak -- the 175.9 MB of SURVEY 8(d) / launch duration -- with the
  plan build reported beside it (it runs once per calibration);
* ``cpu_baseline``: the torch-CPU oracle restatement of the same forward (oracle/torch_model.py)
  timed on this box's host cores on a bounded sample (rank 0, N=1 only), plus the C restatement of the
  voxel-pooling operator single-threaded and with OpenMP;
* ``parity``: the outputs of the very model that was timed against that oracle forward on the same
  frame and weights (max |hip - oracle| over all prediction maps, voxel indices bit-exact); the run
  exits non-zero if the fp32 line is above 1e-3 (bf16 mode: 2e-2 of the output scale);
* ``train_step``: the mixed-precision training step as one hipGraph (cfg-2 model at batch 2; BASELINE configs[3]'s per-GPU share
  through a 1-rank RCCL group), child runs of tools/train_bench.py;
* ``other_configs`` (default cfg-2 run at N=1 only): BASELINE configs[2] / [4] in bf16 as compact records
  from child runs of this script (value, conv-family and voxel-pooling fractions, parity).
"""
import argparse
import json
import os
import sys
import time

# RCCL / cross-process device-memory sharing on this pool needs the dmabuf IPC path (see the image notes);
# exported by the environment normally -- set here as well, before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (--dtype bf16 only)
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step (cfg-2: 1)")
    ap.add_argument("--no-graph", action="store_true", help="eager stream launches instead of a hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-harness-b8", action="store_true", help="skip the batch-8 run of the harness eval_step")
    ap.add_argument("--no-plan-timing", action="store_true",
                    help="roofline_hbm without the plan-build / plan-check timings (keeps a rocprof trace of this command free "
                         "of the plan kernels those timing loops launch)")
    ap.add_argument("--fuse-lift-splat", action="store_true", help="(the default since round 3) rows of the lifted tensor formed inside the pooling gather")
    ap.add_argument("--no-fuse-lift-splat", action="store_true", help="materialise the [B,N,C] lifted tensor: lift kernel + voxel_pooling operator")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight: consecutive steps are replayed round-robin on this many HIP streams "
                         "(each with its own graph and activation buffers), so kernels of frame i+1 fill the CUs that "
                         "the batch-1 layers of frame i leave idle")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "r101", "cfg3", "cfg5"],
                    help="cfg2 (default, the judged workload): R50 BEVHeight; r101: R101 BEVHeight; "
                         "cfg3: R101 1088x1920 -> 512x512 BEV (geometry of BASELINE configs[2]; fp32 here, use --batch 4); "
                         "cfg5: SGV3D BSM R101 (model of BASELINE configs[4], fp32, batch 1)")
    ap.add_argument("--no-harness", action="store_true",
                    help="skip long_run_value and harness_eval_step (keeps a rocprof trace / PMC pass of this command to the timed "
                         "loop's kernels)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact cfg-3 / cfg-5 bf16 records (other_configs) the default cfg-2 run appends")
    ap.add_argument("--sub", action="store_true",
                    help="internal: this is one of the other_configs child runs (prints a compact record)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="warm-up + the instrumented roofline pass only (one stream, eager launches): the command whose rocprofv3 "
                         "--kernel-trace --stats summary is committed as profiles/r*_bench_roofline_kernel_stats.csv -- the per-symbol "
                         "averages there are over exactly the launches roofline.frac is made of")
    ap.add_argument("--no-kernel-timestamps", action="store_true",
                    help="roofline from HIP events only (no torch.profiler / roctracer pass: for runs under rocprofv3, which owns the tracer)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the train_step child records (tools/train_bench.py)")
    ap.add_argument("--no-native-f32", action="store_true",
                    help="skip native_f32_value (a child run with SGV3D_WINO4_X3=0: every product on the f32 MFMA)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32x3", "f32x3auto"],
                    help="f32 (default; what BASELINE cfg-2, the judged line, asks for) or bf16: convolutions multiply on "
                         "the bf16 matrix cores with f32 accumulation (the compute dtype of BASELINE configs[2] / [4]; "
                         "use with --config cfg3 --batch 4 or --config cfg5); f32x3: float32-accurate products from three "
                         "bf16 terms per operand on the bf16 matrix cores (experimental).  Never the default.")
    return ap.parse_args()


def _vp_symbol_form(sym):
    """(kernel family, fused?) of a gather kernel symbol of the PMC summaries, or None: vp_gather_fast_kernel<FB, OB, ACC, FUSED>
    (slot-balanced, round 3) / vp_gather_vox_kernel<FB, OB, ACC, FUSED, VB> (voxel-owner, round 4)."""
    for fam, pos in (("vp_gather_fast_kernel", 3), ("vp_gather_vox_kernel", 3)):
        if sym.startswith(fam + "<"):
            args = [a.strip() for a in sym[len(fam) + 1:].rstrip(">").split(",")]
            return fam, (len(args) > pos and args[pos] == "true")
    return None


def load_vp_traffic(fused=False, family=None):
    """HBM bytes per launch of the voxel-pooling gather from the committed PMC summary: the OPERATOR form (section ``vp_probe``
    = tools/vp_probe.py under the counters) or, with ``fused``, the fused lift-splat form the model launches (section
    ``bench``), of the kernel family that runs here (``family``).  Each figure is attached only to the kernel it was counted
    on; no summary for that kernel -> (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    for f in reversed(files):
        try:
            doc = json.load(open(f))
        except Exception:
            continue
        for section in (("bench",) if fused else ("vp_probe", "bench")):
            for sym, r in (doc.get(section) or {}).items():
                form = _vp_symbol_form(sym)
                if form is not None and form[1] == fused and (family is None or form[0] == family):
                    return r["hbm_bytes_per_launch"], os.path.relpath(f, ROOT) + ":" + section + ":" + sym
    return None, None


def load_traffic(tile_name, symbol=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command, tools/profile_round.sh; FETCH_SIZE doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950).  bench.py cannot read PMC counters itself.

    ``symbol``: the exact kernel symbol the instrumented pass recorded for this label (hip_ops prof ``extra``) -- the
    figure is attached to THAT instantiation only (round 4 prefix-matched ``conv_igemm_kernel<1, 1, true`` and reported
    another instantiation's bytes).  Without one, labels of single-symbol kernels are looked up by their kernel name."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    by_label = {"conv_wino": "conv_wino_kernel", "conv_wino_resident": "conv_wino_resident_kernel",
                "conv_wino_head": "conv_wino_head_kernel", "conv_wino4_resident": "conv_f4res_kernel",
                "conv_head_wino4": "head_wino4_kernel", "conv_head_bf16": "head_bf16_kernel",
                "conv_patch_bf16": "conv_patch_bf16_kernel", "conv_dw_bf16": "conv_dw_bf16_kernel"}
    try:
        table = json.load(open(files[-1]))["bench"]
    except Exception:
        return None, None
    if symbol is not None:
        full = symbol if symbol in table else None
    elif tile_name in by_label:
        cands = [k for k in table if k == by_label[tile_name] or k.startswith(by_label[tile_name] + "<")]
        full = cands[0] if len(cands) == 1 else None          # (several instantiations and no exact symbol: no figure)
    else:
        full = None
    if full is None:
        return None, None
    return table[full]["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT) + ":" + full


def load_rocprof_avg(symbol):
    """(average ns, calls, file) of ``symbol`` in the newest committed ``profiles/r*_bench_kernel_stats.csv`` (rocprofv3
    --kernel-trace --stats of this command), exact template arguments; None when there is no such row."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
    if not files or symbol is None:
        return None
    try:
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if symbol + "(" in row["Name"]:
                    return float(row["AverageNs"]), int(row["Calls"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        pass
    return None


def load_rocprof_roofline(symbol):
    """The newest committed ``profiles/r*_bench_roofline_kernel_stats.csv`` (rocprofv3 --kernel-trace --stats of
    ``bench.py --roofline-only``) with the line that run printed beside it (``..._under_rocprof.json``: steps, warm-up): average
    duration of ``symbol`` and its launches per step there.  None without such a pair."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_roofline_kernel_stats.csv")))
    if not files or symbol is None:
        return None
    try:
        line = json.loads(open(files[-1].replace("_kernel_stats.csv", "_under_rocprof.json")).read().strip().splitlines()[-1])
        rl = line["roofline"]
        if rl["kernel_symbol"] != symbol:
            return None
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if symbol + "(" in row["Name"]:
                    return {"avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"]), "file": os.path.relpath(files[-1], ROOT),
                            "launches_per_step": float(rl["launches_per_step_with_event_pair"]),
                            "flop_per_launch": float(rl["executed_flop_per_launch_all_launches"])}
    except Exception:
        pass
    return None


def run_other_configs(args, budget_s=120.0):
    """cfg-3 (R101 1088x1920 -> 512x512 BEV, batch 4) and cfg-5 (SGV3D BSM R101, batch 1) in bf16 -- the dtype BASELINE
    configs[2] / [4] name -- as child runs of this script: value, ms, conv-family fraction of the bf16 MFMA peak, voxel
    pooling against HBM peak and the parity of the timed model against the oracle (2e-2 of the output scale).  One GPU
    process at a time; the children reuse the committed tune DB (tune/), so they spend no time on candidate timing."""
    import subprocess
    out, t_start = [], time.perf_counter()
    # (timed steps sized for a ~0.15 s timed region each: ten 3-ms steps of cfg-5 read 8 % low against a longer run -- clocks and
    # the first graph replays)
    for cfg, batch, steps, warm in (("cfg3", 4, 12, 3), ("cfg5", 1, 40, 5)):
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20.0:
            out.append({"config": cfg, "skipped": f"other_configs budget of {budget_s:.0f} s spent"})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--sub", "--config", cfg, "--batch", str(batch), "--dtype", "bf16",
               "--steps", str(steps), "--warmup", str(warm), "--streams", str(args.streams), "--no-plan-timing"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left + 30.0)
            rec = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:          # a failed child is reported, never silently dropped
            out.append({"config": cfg, "error": repr(e)[:300]})
            continue
        rf, rh, par = rec.get("roofline") or {}, rec.get("roofline_hbm") or {}, rec.get("parity") or {}
        out.append({"config": cfg, "dtype": "bf16", "batch_per_gpu": batch, "workload": rec["config"]["workload"],
                    "value": rec["value"], "unit": "frames/s", "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
                    "frames_in_flight": rec["config"]["frames_in_flight"],
                    "conv_family_frac": (rf.get("conv_family") or {}).get("frac"), "dominant_kernel": rf.get("kernel"),
                    # against BOTH roofs per launch (the large 1x1 layers of these configs are HBM-bound in bf16)
                    "conv_family_two_roof_frac": (rf.get("conv_family") or {}).get("two_roof_frac"),
                    "conv_family_hbm_bound_launches_per_step": (rf.get("conv_family") or {}).get("hbm_bound_launches_per_step"),
                    "dominant_kernel_frac": rf.get("frac"), "dominant_kernel_two_roof_frac": (rf.get("two_roof") or {}).get("frac"),
                    "roofline_hbm_frac": rh.get("frac"),
                    "parity": {k: par.get(k) for k in ("max_abs_err", "max_abs_ref", "rel_err", "tolerance",
                                                       "voxel_indices_equal", "ok")},
                    "exit_code": r.returncode, "wall_s": time.perf_counter() - t0})
    return out


def run_train_steps(budget_s=150.0):
    """The training step (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:224-240,298-305,405) under this
    script's clock, as child runs of tools/train_bench.py: the cfg-2 model at batch 2 and at BASELINE configs[3]'s per-GPU share
    (batch 4, through a 1-rank RCCL group: broadcast, bucket all-reduces from inside backward, clipped fused AdamW), both as
    mixed-precision steps (bf16 products, f32 master weights) recorded as one hipGraph, frozen stem and gradient_clip_val=5 as the
    reference's Trainer has them.  Compact records; a failed child is reported, not dropped."""
    import socket
    import subprocess
    out, t_start = [], time.perf_counter()
    base = {k: v for k, v in os.environ.items() if k not in ("SGV3D_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    tb = os.path.join(ROOT, "tools", "train_bench.py")
    for tag, extra, dist in (("cfg2_batch2", ["--config", "cfg2", "--batch", "2"], False), ("cfg4_share_batch4_1rank_rccl", ["--config", "cfg4"], True)):
        left = budget_s - (time.perf_counter() - t_start)
        if left < 25.0:
            out.append({"case": tag, "skipped": f"train_step budget of {budget_s:.0f} s spent"})
            continue
        env = dict(base)
        if dist:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            env.update(SGV3D_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, tb, "--dtype", "bf16", "--graph", "--steps", "10", "--warmup", "2"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left + 30.0)
            rec = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            out.append({"case": tag, "error": repr(e)[:300]})
            continue
        keep = ("ms_per_step", "value", "unit", "batch_per_gpu", "dtype", "graph", "update_in_graph", "steps", "loss", "backend", "world_size",
                "collectives_active", "allreduce_buckets", "allreduce_bytes_per_step", "allreduces_launched_inside_backward",
                "first_allreduce_launch_at_fraction_of_backward", "bucket_mib", "gradient_clip_val", "grad_norm_last_step",
                "clip_coefficient_last_step", "frozen_parameters", "peak_mem_gb", "parameters")
        out.append(dict({"case": tag, "command": "tools/train_bench.py " + " ".join(cmd[2:]), "exit_code": r.returncode,
                         "wall_s": time.perf_counter() - t0}, **{k: rec.get(k) for k in keep}))
    return out


def run_native_f32(args, budget_s=60.0):
    """``value`` of this very command with every product on the f32 MFMA (SGV3D_WINO4_X3=0 SGV3D_PW_X3=0: the F(4x4) position GEMMs
    and the 1x1 layers on v_mfma_f32_* instead of six bf16 partial products per f32 product) -- a child run, timed region only."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sub", "--config", args.config, "--batch", str(args.batch), "--dtype", "f32",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--streams", str(args.streams), "--no-roofline", "--no-cpu-baseline"]
    try:
        # (no SGV3D_TUNE_CACHE in the child: it re-measures the f32x3 layers among the f32 candidates and must not write that back)
        env = {k: v for k, v in os.environ.items() if k != "SGV3D_TUNE_CACHE"}
        env.update(SGV3D_WINO4_X3="0", SGV3D_PW_X3="0")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s + 30.0)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"], "exit_code": r.returncode,
                "layers_measured_here": rec["config"]["per_rank"][0]["layers_measured_here"],
                "what": "child run of this command with SGV3D_WINO4_X3=0 SGV3D_PW_X3=0; the layers whose committed choice is an f32x3 tile are "
                        "re-measured among the f32-MFMA candidates at its first forward (layers_measured_here)"}
    except Exception as e:
        return {"error": repr(e)[:300]}


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, one
    fresh child process per GPU with the environment ``torch.distributed.run`` would give it (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT on 127.0.0.1), forward rank 0's single JSON line and exit with the worst child's code.  This
    parent never touches the GPU (it only counts devices, which does not initialise HIP), and nothing is exec'd: the ranks
    are ordinary children.  Fewer than N visible devices is an error, not a silent one-rank run."""
    import socket
    import subprocess
    n = args.gpus
    stub = bool(os.environ.get("SGV3D_BENCH_STUB"))
    if not stub:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible and no launcher environment (WORLD_SIZE unset); "
                             f"refusing to run fewer ranks than asked for")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    worst = max((c for c in codes), key=abs)
    if worst:
        print(f"[bench] rank exit codes: {codes}", file=sys.stderr)
    sys.exit(worst if worst >= 0 else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)           # never returns
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    # stdout carries exactly ONE line, the JSON record of rank 0.  Libraries write there too (RCCL prints a
    # version banner through C stdio, which would surface after our line at exit), so file descriptor 1 points
    # to stderr for the whole run and is restored only for the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    # SGV3D_BENCH_STUB=1 (tests/test_multigpu_cpu.py only): a CPU rehearsal of THIS script's multi-rank protocol -- process
    # group, barriers, MAX over ranks, gather of the per-rank records, final barrier, one JSON line from rank 0 -- with the
    # model replaced by a sleep.  It measures nothing and says so in the line ("data": "stub").
    stub = bool(os.environ.get("SGV3D_BENCH_STUB"))
    if stub:
        args.no_roofline = args.no_cpu_baseline = True
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    if not stub:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    from sgv3d_amd import hip_ops, synthetic as S
    from sgv3d_amd.replicas import ReplicaGroup
    group = ReplicaGroup(backend=("gloo" if stub else "nccl") if (world > 1 or os.environ.get("SGV3D_FORCE_DIST")) else None,
                         device=None if stub else dev)                              # nccl == RCCL on ROCm
    from sgv3d_amd.models.bev_height import BEVHeight
    hip_ops.MFMA_BF16 = args.dtype == "bf16"
    hip_ops.MFMA_F32X3 = True if args.dtype == "f32x3" else ("auto" if args.dtype == "f32x3auto" else False)
    peak = MFMA_BF16_PEAK_TFLOPS if hip_ops.MFMA_BF16 else MFMA_F32_PEAK_TFLOPS

    bc, hc = {"cfg2": S.r50_256_conf, "r101": S.r101_256_conf, "cfg3": S.r101_512_conf,
              "cfg5": S.bsm_r101_256_conf}[args.config]()
    workload = {"cfg2": "BASELINE cfg-2: ResNet-50 864x1536 -> 256x256 BEV, fp32, full BEVHeight forward "
                        "(backbone+neck+HeightNet+lift+voxel_pooling+head); the calibration arrives as fresh tensor objects with every "
                        "frame, as the reference harness hands it over: copy + geometry kernel + device-side compare of the voxel indices "
                        "per frame inside the timed region, the voxel plan and the camera gates rebuilt only when the indices changed "
                        "(cached_calibration_value: the caller keeps its calibration tensors)",
                "r101": "ResNet-101 864x1536 -> 256x256 BEV, fp32, full BEVHeight forward",
                "cfg3": "ResNet-101 1088x1920 (1080 padded) -> 512x512 BEV, fp32 (BASELINE configs[2] asks bf16), "
                        "full BEVHeight forward",
                "cfg5": "SGV3D BSM ResNet-101 864x1536 -> 256x256 BEV (stride-8 frustum, D=180, 87-ch BEV), fp32, "
                        "full forward"}[args.config]
    if args.dtype == "f32x3auto":
        workload = workload.replace("fp32", "fp32; per layer the faster of the f32 MFMA and 3 x bf16 split operands on the bf16 MFMA (f32-accurate)")
    if args.dtype == "f32x3":
        workload = workload.replace("fp32", "fp32 products as 3 x bf16 split operands on the bf16 MFMA, f32 accumulation")
    if args.dtype == "bf16":
        workload = workload.replace("fp32 (BASELINE configs[2] asks bf16)", "fp32").replace(
            "fp32", "bf16 MFMA operands / f32 accumulation, bf16 activations in HBM inside the conv chains "
                    "(height / depth logits, context, lifted features, BEV map and predictions f32)")
    B = args.batch
    if args.roofline_only:
        # one stream, eager launches from the first to the last forward of the process: every launch of a kernel symbol in a
        # rocprofv3 trace of this command is then of the population roofline.frac is made of (kernels alone on the chip)
        args.streams, args.no_graph, args.no_cpu_baseline, args.no_kernel_timestamps = 1, True, True, True
        args.no_harness = args.no_other_configs = args.no_train_step = args.no_native_f32 = args.no_plan_timing = True
    nstreams = max(1, args.streams)
    if stub:
        class _StubPipe:
            caches, use_graph = [], False

            def submit(self, imgs, mats):
                time.sleep(0.002 * (1 + rank))           # rank-dependent: the MAX over ranks has to pick the slowest rank
        model = imgs = mats = None
        pipe, use_graph = _StubPipe(), False
        workload = "STUB (SGV3D_BENCH_STUB=1): control-flow rehearsal on the CPU, no model -- not a measurement"
    else:
        torch.manual_seed(0)
        model = BEVHeight(bc, hc).eval()
        S.randomize_norm_stats_(model, 0, residual_gamma=0.3)    # small last-BN gamma per residual block, as mmdet initialises
        model = model.to(dev)
        model.backbone.fuse_lift_splat = not args.no_fuse_lift_splat
        imgs = S.make_images(B, bc['final_dim'], device=dev, seed=rank)
        mats = S.make_mats(B, device=dev)

    def step():
        with torch.no_grad():
            return model(imgs, mats)

    if not stub:
        # ---- warm-up: packs weights, tunes tiles, fills the caching allocator ---------------------
        # (the per-layer candidates are timed as isolated launches, hip_ops.TUNE_STREAMS = 1: the committed tune DB's "|ts1" entries)
        from sgv3d_amd.pipeline import eager_forward
        with eager_forward(model):      # (plain launches: BEVHeight's own per-signature graph is what harness_eval_step measures)
            for _ in range(max(1, args.warmup)):
                out = step()
        torch.cuda.synchronize()
        hip_ops.save_tune_db()          # no-op unless SGV3D_TUNE_CACHE is set (tools/profile_round.sh)

        # ---- hipGraph capture: one graph + activation pool per frame in flight (sgv3d_amd/pipeline.py) ----
        from sgv3d_amd.pipeline import FramePipeline
        pipe = FramePipeline(model, imgs, mats, slots=nstreams, use_graph=not args.no_graph)
        use_graph = pipe.use_graph
        if not use_graph and not args.no_graph and rank == 0:
            print("[bench] hipGraph capture failed; running eager launches on the slot streams", file=sys.stderr)
    # The product call: resident frame -> slot's static inputs -> graph replay.  The calibration arrives as NEW tensor objects with
    # every frame, as the reference harness hands it over (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:244-247
    # makes fresh `mats[k].cuda()` tensors every step): the host fast path of the calibration cache never fires, every submit copies
    # the seven tensors into the slot, re-runs calib_prep + the geometry kernel and the device-side compare of the voxel indices
    # (the plan itself is rebuilt only when the indices really changed).  A ring of distinct tensor objects with the same content.
    ring = None if stub else [{k: v.clone() for k, v in mats.items()} for _ in range(2 * nstreams + 1)]
    ctr = [0]

    def run():
        ctr[0] += 1
        return pipe.submit(imgs, mats if stub else ring[ctr[0] % len(ring)])
    for _ in range(args.warmup * nstreams):
        run()
    if not stub:
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps ------------------------------------------------------------
    elapsed_local = [0.0]
    counters0 = (sum(c.hits for c in pipe.caches), sum(c.refreshes for c in pipe.caches),
                 sum(c.plan.builds() for c in pipe.caches if c.plan is not None))

    def timed_local(fn, n):
        t = group.timed(fn, n, local_out=elapsed_local)
        return t
    if args.roofline_only:
        args.steps = max(1, args.steps)
    elapsed = timed_local(run, args.steps)          # barrier+sync | K steps | barrier+sync, MAX over ranks
    value = group.aggregate_throughput(B, args.steps, elapsed)
    # calibration cache counters of the timed slots: geometry / plan kernels launched inside the timed region
    calib = {"plan_builds_before_timed_region": counters0[2],
             "plan_builds_in_timed_region": sum(c.plan.builds() for c in pipe.caches if c.plan is not None) - counters0[2],
             "geometry_launches_in_timed_region": sum(c.refreshes for c in pipe.caches) - counters0[1],
             "calibration_tensors": "fresh objects every frame (the reference harness's call pattern)"}
    quick = args.roofline_only          # only the warm-up, the K steps above and the instrumented pass
    single = None
    if nstreams > 1 and rank == 0 and world == 1 and not args.sub and not stub and not quick:   # same K steps with one frame in flight, for reference
        one = FramePipeline(model, imgs, mats, slots=1, use_graph=use_graph)
        def run_one():
            ctr[0] += 1
            return one.submit(imgs, ring[ctr[0] % len(ring)])
        for _ in range(args.warmup):
            run_one()
        t1 = group.timed(run_one, args.steps)
        single = {"value": B * args.steps / t1, "ms_per_step": t1 / args.steps * 1e3}
        del one
    # ---- the same K steps with the calibration tensors of the warm-up handed in again every frame (a caller that keeps its
    # calibration tensors: nothing of the view transform's setup is launched) -- rounds 1-5 reported THIS figure as `value`
    cached = None
    if rank == 0 and world == 1 and not args.sub and not stub and not quick:
        for _ in range(args.warmup * nstreams):
            pipe.submit(imgs, mats)
        c1 = (sum(c.refreshes for c in pipe.caches), sum(c.plan.builds() for c in pipe.caches if c.plan is not None))
        tf = group.timed(lambda: pipe.submit(imgs, mats), args.steps)
        cached = {"value": B * args.steps / tf, "ms_per_step": tf / args.steps * 1e3,
                 "geometry_launches_in_timed_region": sum(c.refreshes for c in pipe.caches) - c1[0],
                 "plan_builds_in_timed_region": sum(c.plan.builds() for c in pipe.caches if c.plan is not None) - c1[1]}
    # ---- ... and ANOTHER calibration with every frame (a stream that interleaves cameras, as a shuffled DAIR-V2X-I validation
    # set does): the voxel indices really change, so every submit rebuilds the slot's plan (~16 launches, 119 us) and the
    # height net's camera gates on top of the geometry kernel.  Same K steps; two calibrations in turn per slot.
    changing = None
    if rank == 0 and world == 1 and not args.sub and not stub and B == 1 and not quick:
        other = {k: v[1:2].clone() for k, v in S.make_mats(2, device=dev).items()}      # sample 1 of a varied pair: another camera
        pair = [mats, other]
        ctr2 = [0]

        def run_changing():
            ctr2[0] += 1
            return pipe.submit(imgs, pair[(ctr2[0] // nstreams) % 2])       # (each slot sees the two calibrations alternate)
        for _ in range(args.warmup * nstreams):
            run_changing()
        c2 = (sum(c.refreshes for c in pipe.caches), sum(c.plan.builds() for c in pipe.caches if c.plan is not None))
        tc = group.timed(run_changing, args.steps)
        changing = {"value": B * args.steps / tc, "ms_per_step": tc / args.steps * 1e3,
                    "geometry_launches_in_timed_region": sum(c.refreshes for c in pipe.caches) - c2[0],
                    "plan_builds_in_timed_region": sum(c.plan.builds() for c in pipe.caches if c.plan is not None) - c2[1]}
        for _ in range(2 * nstreams):               # leave every slot on the benchmark's own calibration again
            pipe.submit(imgs, mats)
        torch.cuda.synchronize()
    # ---- a timed region of >= 1 s with the same pipeline (the K-step region above is ~0.1 s at cfg-2: clock ramp and
    # the first replays weigh on it; `value` stays the K-step figure the contract asks for, this one sits beside it)
    long_run = None
    if rank == 0 and world == 1 and not args.sub and not stub and not args.no_harness and not quick:
        n_long = max(args.steps, int(1.2 / max(elapsed / args.steps, 1e-4)) + 1)
        tl = group.timed(run, n_long)
        long_run = {"value": B * n_long / tl, "ms_per_step": tl / n_long * 1e3, "steps": n_long, "seconds": tl}
    # every rank's own record: its frames/s, the device it was bound to, and whether its layers were answered from the
    # committed tune DB (tune/gfx950_*.json, loaded by every process at import) or timed by this rank -- with eight ranks on
    # one host a first-call measurement competes for the host cores, so `layers_measured_here` should be 0 on an MI355X
    per_rank = group.all_gather_object({"rank": rank, "local_rank": local_rank, "frames_per_s": B * args.steps / elapsed_local[0],
                                        "device": "cpu (stub)" if stub else torch.cuda.get_device_name(dev),
                                        "device_index": None if stub else torch.cuda.current_device(),
                                        "tune_db_entries": len(hip_ops.TUNE_DB),
                                        "layers_from_tune_db": hip_ops.TUNE_STATS["from_db"],
                                        "layers_measured_here": hip_ops.TUNE_STATS["measured"]})

    # ---- roofline: instrumented passes over the same K steps (one stream, eager launches) -----------------------
    # pass A: HIP events around every launch (hip_ops.PROFILE: label, algorithmic flops, bytes, kernel symbol, executed flops);
    # pass B: the same K steps under torch.profiler = roctracer's kernel begin / end timestamps, the quantity a rocprofv3
    #         --kernel-trace of this loop reports per kernel symbol.  An event pair also measures the gap its two markers take on the
    #         queue (~6 us on a 30 us launch), so the durations roofline.frac is made of are pass B's; pass A's -- raw and with the
    #         calibrated gap subtracted -- sit beside them with their ratio (roofline.hip_events).
    roofline = None
    if rank == 0 and not args.no_roofline:
        hip_ops.PROFILE = []
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        recs = hip_ops.PROFILE
        hip_ops.PROFILE = None
        kernel_ts, kernel_ts_error = None, None
        # (under rocprofv3 a second tracer in the process -- torch.profiler -- delivers durations of ~1 us for 25-us kernels: the
        #  profiler's preloaded tool owns the queue's timestamps.  The trace itself is then the timestamp source.)
        under_rocprof = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if under_rocprof and not args.no_kernel_timestamps:
            kernel_ts_error = "running under rocprofv3: its own trace holds the kernel timestamps"
        if not args.no_kernel_timestamps and not under_rocprof:
            try:
                from torch.profiler import profile as _tprofile, ProfilerActivity as _Act
                from torch.autograd import DeviceType as _DevT
                # (a session also delivers activity records of launches that preceded it -- graph replays of the timed loops --, so
                #  pass B starts behind a marker kernel and only what began after the marker's end is counted)
                from sgv3d_amd.pipeline import eager_forward as _eager
                with _tprofile(activities=[_Act.CUDA], acc_events=True) as _prof, _eager(model):     # (plain launches, as in pass A)
                    torch.cuda._sleep(20000)
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                evs = [e for e in _prof.events() if e.device_type == _DevT.CUDA]
                marks = [e for e in evs if "sleep" in e.name.lower() or "spin" in e.name.lower()]
                if not marks:
                    raise RuntimeError("marker kernel not in the trace: " + ", ".join(sorted({e.name[:40] for e in evs})[:6]))
                t_mark = max(e.time_range.end for e in marks)
                kernel_ts = {}
                for e in evs:
                    if e.time_range.start >= t_mark:
                        q = kernel_ts.setdefault(e.name.replace(" ", ""), [0, 0.0])
                        q[0] += 1
                        q[1] += float(e.device_time) * 1e-6           # us -> s
                del _prof
            except Exception as ex:                  # reported in the line; the HIP-event figure then stands alone
                kernel_ts, kernel_ts_error = None, repr(ex)[:200]

        def ts_of(symbol):
            """(launches, seconds) of a kernel symbol in pass B, or None."""
            if kernel_ts is None or symbol is None:
                return None
            key = symbol.replace(" ", "") + "("
            hit = [v for k, v in kernel_ts.items() if key in k]
            return (sum(v[0] for v in hit), sum(v[1] for v in hit)) if hit else None
        cal = []
        for _ in range(200):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); c1.record()
            cal.append((c0, c1))
        torch.cuda.synchronize()
        gap_s = sorted(c0.elapsed_time(c1) for c0, c1 in cal)[len(cal) // 2] * 1e-3
        by_kernel, by_symbol, raw_sec, sym_ev = {}, {}, {}, {}
        HBM_PEAK = 8.0e12
        # the three-launch F(4x4) labels hold two transform kernels beside the grouped GEMMs: no event pair of a kernel's own there
        THREE_LAUNCH = ("conv_wino4", "conv_wino4_x3")
        def executed_of(name, flops):
            return flops / 4.0 if "wino4" in name else flops / 2.25 if "wino" in name else flops
        for name, flops, e0, e1, nbytes, extra in recs:
            d = by_kernel.setdefault(name, [0.0, 0.0, 0, 0.0, {}, 0.0, 0])
            raw_s = e0.elapsed_time(e1) * 1e-3
            cor_s = max(raw_s - gap_s, 1e-7)
            d[0] += flops
            d[1] += cor_s
            d[2] += 1
            d[3] += nbytes
            # the launch's floor under BOTH roofs: executed flops at the MFMA peak of the products it runs (an f32x3 kernel: its six
            # bf16 partial products per f32 product at the bf16 peak), algorithmic bytes at the HBM peak
            bf16_fl = extra.get("bf16_mfma_flops", 0.0) if extra else 0.0
            t_mfma = bf16_fl / (MFMA_BF16_PEAK_TFLOPS * 1e12) if bf16_fl else executed_of(name, flops) / (peak * 1e12)
            t_hbm = nbytes / HBM_PEAK
            d[5] += max(t_mfma, t_hbm)
            d[6] += 1 if t_hbm > t_mfma else 0
            raw_sec[name] = raw_sec.get(name, 0.0) + raw_s
            if extra:
                d[4][extra["symbol"]] = d[4].get(extra["symbol"], 0) + 1
                q = by_symbol.setdefault(extra["symbol"], [0.0, 0, 0.0])
                q[0] += extra["mfma_flops"]
                q[1] += 1
                q[2] += bf16_fl
                if name.startswith("conv_") and name not in THREE_LAUNCH:
                    # one event pair = one launch of this symbol: the per-SYMBOL HIP-event record (a label such as conv_pw_x3 covers
                    # several instantiations, so labels are not the unit here)
                    v = sym_ev.setdefault(extra["symbol"], {"f32": 0.0, "bf16": 0.0, "n": 0, "raw": 0.0, "cor": 0.0, "bytes": 0.0,
                                                            "alg": 0.0, "floor": 0.0, "hbm_bound": 0, "labels": {}})
                    v["f32"] += extra["mfma_flops"]; v["bf16"] += bf16_fl; v["n"] += 1; v["raw"] += raw_s; v["cor"] += cor_s
                    v["bytes"] += nbytes; v["alg"] += flops; v["floor"] += max(t_mfma, t_hbm); v["hbm_bound"] += 1 if t_hbm > t_mfma else 0
                    v["labels"][name] = v["labels"].get(name, 0) + 1
        # `flops` of a record is the ALGORITHMIC work of the layer (2 x MACs of the direct convolution,
        # SURVEY 8d).  The Winograd F(2x2,3x3) kernels execute 1/2.25 of it on the MFMA pipe, so their
        # algorithmic rate can exceed the hardware peak; the executed rate is reported beside it.
        def executed(name, flops):
            return executed_of(name, flops)
        conv = {k: v for k, v in by_kernel.items() if k.startswith("conv_")}
        fam_fl = sum(v[0] for v in conv.values())
        fam_ex = sum(executed(k, v[0]) for k, v in conv.items())
        fam_sec = sum(v[1] for v in conv.values())
        all_sec = sum(v[1] for v in by_kernel.values())
        # ---- the dominant KERNEL = the kernel SYMBOL with the largest summed (gap-corrected) HIP-event time among the launches that
        # have an event pair of their own (the grouped GEMMs inside a three-launch F(4x4) label are in `by_symbol`, from the kernel
        # timestamps).  An f32x3 kernel runs bf16 MFMAs: its achieved rate is the bf16 flops it EXECUTES (six partial products per f32
        # product) and its peak the dense bf16 peak -- never the f32-equivalent rate over the f32 peak.
        if not sym_ev:
            # no launch of this configuration records its kernel symbol (the bf16 kernels of --dtype bf16: one instantiation per
            # label): the dominant LABEL stands for its kernel, flops = executed flops of the label
            one_kernel = {k: v for k, v in conv.items() if k not in THREE_LAUNCH} or conv
            lab = max(one_kernel, key=lambda k: one_kernel[k][1])
            v = conv[lab]
            sym_ev[None] = {"f32": executed(lab, v[0]), "bf16": 0.0, "n": v[2], "raw": raw_sec[lab], "cor": v[1], "bytes": v[3],
                            "alg": v[0], "floor": v[5], "hbm_bound": v[6], "labels": {lab: v[2]}}
            by_symbol[None] = [executed(lab, v[0]), v[2], 0.0]
        top_symbol = max(sym_ev, key=lambda k: sym_ev[k]["cor"])
        ev = sym_ev[top_symbol]
        top = max(ev["labels"], key=ev["labels"].get)
        is_x3 = ev["bf16"] > 0.0
        k_peak = MFMA_BF16_PEAK_TFLOPS if is_x3 else peak
        ev_fl = ev["bf16"] if is_x3 else ev["f32"]                     # executed flops of the event-pair launches, in the kernel's products
        fl, sec, n, nby, floor_sec, hbm_bound_n = ev["alg"], ev["cor"], ev["n"], ev["bytes"], ev["floor"], ev["hbm_bound"]
        traffic, traffic_src = load_traffic(top, top_symbol)
        # every launch of the symbol in the pass (for the five-per-CU pointwise tile also grouped GEMMs of F(4x4) layers that use the
        # same instantiation): executed flops over kernel-timestamp durations when pass B saw the same launches
        all_f32, all_n, all_bf16 = by_symbol[top_symbol]
        ts = ts_of(top_symbol)
        if os.environ.get("SGV3D_BENCH_DEBUG"):
            print("[bench debug] by_symbol", {k: v[1] for k, v in by_symbol.items()}, file=sys.stderr)
            print("[bench debug] labels", {k: (v[2], dict(v[4])) for k, v in by_kernel.items()}, file=sys.stderr)
            if kernel_ts:
                print("[bench debug] kernel_ts", {k[:110]: v[0] for k, v in kernel_ts.items()}, file=sys.stderr)
        ts_ok = ts is not None and ts[0] == all_n                     # the same population in both passes, or no figure
        if ts_ok and not (0.6 < (ts[1] / ts[0]) / (ev["raw"] / n) < 1.2):
            # kernel timestamps that disagree grossly with the event pairs around the same launches are a broken trace, not a fast kernel
            kernel_ts = None                                          # (by_symbol stays empty too)
            ts_ok, kernel_ts_error = False, f"timestamps implausible: {ts[1] / ts[0] * 1e6:.2f} us against {ev['raw'] / n * 1e6:.2f} us between events"
        if ts_ok:
            sym_fl, sym_n, dur_s = (all_bf16 if is_x3 else all_f32), all_n, ts[1]
            dur_src = "kernel begin/end timestamps (torch.profiler = roctracer), pass B"
        else:
            sym_fl, sym_n, dur_s = ev_fl, n, sec
            dur_src = "HIP events, calibrated marker gap subtracted (no kernel timestamps: " + (kernel_ts_error or (
                "--no-kernel-timestamps" if args.no_kernel_timestamps else f"launch counts differ, {ts} vs {all_n}")) + ")"
        achieved = sym_fl / dur_s / 1e12
        rocprof = None
        rp = load_rocprof_roofline(top_symbol)
        if rp is not None:
            # the committed trace of `bench.py --roofline-only` (same loop, same tune DB): refuse another population -- the launches
            # per step WITH an event pair are what both runs print, and the executed flops per launch of the symbol must agree
            fpl = (all_bf16 if is_x3 else all_f32) / all_n
            same = abs(rp["launches_per_step"] - n / args.steps) < 1e-9 and abs(rp["flop_per_launch"] / fpl - 1.0) < 1e-6
            rocprof = {"file": rp["file"], "symbol": top_symbol, "avg_ns": rp["avg_ns"], "calls_in_trace": rp["calls"],
                       "launches_per_step_in_trace": rp["launches_per_step"], "launches_per_step_here": n / args.steps,
                       "executed_flop_per_launch_in_trace": rp["flop_per_launch"], "executed_flop_per_launch_here": fpl,
                       "achieved": (fpl / rp["avg_ns"] / 1e3) if same else None,
                       "frac": (fpl / rp["avg_ns"] / 1e3 / k_peak) if same else None,
                       "refused": None if same else "the trace holds another launch mix of this symbol than this run (tune DB or switches "
                                                    "changed since it was committed): no figure",
                       "what": "executed flops per launch of this symbol in THIS run / the symbol's average duration in the committed "
                               "rocprofv3 --kernel-trace --stats summary of `bench.py --roofline-only` (the same loop)"}
        # every MFMA kernel symbol whose executed flops the pass recorded, from the kernel timestamps
        by_sym_out = {}
        for sym, (f32_fl, cnt, bf16_fl) in by_symbol.items():
            t = ts_of(sym)
            if t is None or t[0] != cnt:
                continue
            rec = {"launches_per_step": cnt / args.steps, "avg_us": t[1] / cnt * 1e6, "ms_per_step": t[1] / args.steps * 1e3,
                   "executed_f32_equivalent_tflops": f32_fl / t[1] / 1e12}
            if bf16_fl:
                # f32x3: six bf16 partial products per f32 product -- priced against the bf16 peak with the bf16 flops it executes
                rec.update({"bound": "mfma bf16", "executed_bf16_tflops": bf16_fl / t[1] / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                            "frac": bf16_fl / t[1] / 1e12 / MFMA_BF16_PEAK_TFLOPS})
            else:
                rec.update({"bound": "mfma", "peak": peak, "frac": f32_fl / t[1] / 1e12 / peak})
            by_sym_out[sym] = rec
        roofline = {
            "bound": "mfma bf16 (f32x3: three bf16 planes per f32 operand, six partial products)" if is_x3 else "mfma",
            "kernel": top, "kernel_symbol": top_symbol, "achieved": achieved, "peak": k_peak,
            "unit": "TFLOP/s", "frac": achieved / k_peak, "traffic": traffic,
            "traffic_source": traffic_src,
            "flops": ("executed bf16 MFMA flops (6 x the f32 products of the implicit GEMM, padded tiles included)" if is_x3
                      else "executed (= algorithmic for an implicit GEMM)"), "durations": dur_src,
            "launches": sym_n, "launches_per_step": sym_n / args.steps, "avg_launch_us": dur_s / sym_n * 1e6,
            "executed_flop_per_launch": sym_fl / sym_n,
            "executed_flop_per_launch_all_launches": (all_bf16 if is_x3 else all_f32) / all_n,
            "launches_per_step_with_event_pair": n / args.steps,
            "f32_equivalent": ({"achieved": all_f32 / all_n * sym_n / dur_s / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                "what": "the f32 products the kernel stands for, per second, beside the native f32 MFMA peak -- not "
                                        "a roofline fraction: the kernel runs on the bf16 pipe"} if is_x3 else None),
            # the contract's HIP-event measurement: the event pairs around the launches of THIS symbol, raw and gap-corrected
            "hip_events": {"launches": n, "avg_launch_us_raw": ev["raw"] / n * 1e6, "avg_launch_us": sec / n * 1e6,
                           "event_pair_gap_us": gap_s * 1e6,
                           "achieved_raw": ev_fl / ev["raw"] / 1e12, "achieved": ev_fl / sec / 1e12,
                           "frac_raw": ev_fl / ev["raw"] / 1e12 / k_peak, "frac": ev_fl / sec / 1e12 / k_peak,
                           "gap_corrected_over_timestamps": ((ev_fl / sec) / (sym_fl / dur_s)) if ts_ok else None,
                           "what": "HIP events on the launch stream around every launch of the symbol that is a label of its own; an "
                                   "event pair also measures its markers' gap on the queue (calibrated on empty pairs, subtracted in "
                                   "the second figure)"},
            "algorithmic_bytes": nby / n if nby else None,
            "traffic_over_algorithmic": (traffic / (nby / n)) if (traffic and nby) else None,
            "frac_from_rocprof": rocprof,
            "kernel_timestamps_error": kernel_ts_error,
            "by_symbol": by_sym_out,
            "algorithm": "winograd F(2x2,3x3): executes 1/2.25 of the algorithmic flops" if "wino" in top else "implicit GEMM",
            "flop_per_launch": fl / n,
            # against BOTH roofs per launch (a 1x1 layer with 64 input channels is HBM-bound, not MFMA-bound): sum over the launches
            # of max(executed flops / MFMA peak, algorithmic bytes / 8 TB/s) over the measured time
            "two_roof": {"frac": floor_sec / sec, "floor_us_per_launch": floor_sec / n * 1e6, "hbm_bound_launches": hbm_bound_n,
                         "hbm_peak_tb_s": 8.0},
            "conv_family": {"achieved": fam_fl / fam_sec / 1e12, "frac": fam_fl / fam_sec / 1e12 / peak,
                            "executed_achieved": fam_ex / fam_sec / 1e12,
                            "executed_frac": fam_ex / fam_sec / 1e12 / peak,
                            "executed_frac_note": ("f32-equivalent products over the native f32 MFMA peak; layers on f32x3 tiles run on "
                                                   "the bf16 pipe (by_symbol prices them against the bf16 peak), so this is the "
                                                   "family's rate in f32 terms, not a fraction of the pipe they occupy"
                                                   if any(v[2] for v in by_symbol.values()) else None),
                            "gflop_per_frame": fam_fl / (args.steps * B) / 1e9,
                            "ms_per_step": fam_sec / args.steps * 1e3,
                            "share_of_instrumented_time": fam_sec / all_sec,
                            "two_roof_frac": sum(v[5] for v in conv.values()) / fam_sec,
                            "hbm_bound_launches_per_step": sum(v[6] for v in conv.values()) / args.steps,
                            "by_kernel": {k: {"ms_per_step": v[1] / args.steps * 1e3,
                                              "achieved": v[0] / v[1] / 1e12,
                                              "executed_achieved": executed(k, v[0]) / v[1] / 1e12,
                                              "launches_per_step": v[2] / args.steps,
                                              "algorithmic_bytes_per_launch": v[3] / v[2] if v[3] else None,
                                              "two_roof_frac": v[5] / v[1],
                                              # (a label that covers several instantiations names no single symbol)
                                              "symbol": (next(iter(v[4])) if len(v[4]) == 1 else None),
                                              "symbols": ({q: c / args.steps for q, c in v[4].items()} if len(v[4]) > 1 else None)}
                                          for k, v in conv.items()}},
            "other_kernels_ms_per_step": {k: v[1] / args.steps * 1e3 for k, v in by_kernel.items()
                                          if not k.startswith("conv_")},
            "method": "two instrumented passes over the same K eager steps on one stream: HIP events around every launch (conv_family, "
                      "two_roof, hip_events) and roctracer kernel timestamps through torch.profiler (achieved / frac, by_symbol)",
            "event_pair_gap_us": gap_s * 1e6,
        }
        # traffic / algorithmic bytes of every conv kernel whose symbol has a row in the PMC summary
        for k, rec in roofline["conv_family"]["by_kernel"].items():
            tr, src = load_traffic(k, rec["symbol"])
            if tr is not None and rec["algorithmic_bytes_per_launch"]:
                rec["traffic"], rec["traffic_source"] = tr, src
                rec["traffic_over_algorithmic"] = tr / rec["algorithmic_bytes_per_launch"]

    # ---- voxel pooling against the HBM roofline (north_star: >= 60 % of HBM peak) -----------------
    roofline_hbm = None
    if rank == 0 and not args.no_roofline and not quick:
        from sgv3d_amd.ops.voxel_pooling import VoxelPlan
        bb = model.backbone
        with torch.no_grad():
            geom, plan = bb.calibration(mats, 0)
        Bn, Np = plan.B, plan.N
        Cvp = bb.output_channels if not bc.get('is_bsm') else (bb.bev_channels + 3) // 4 * 4
        feats = torch.randn(Bn, Np, Cvp, device=dev)
        X, Y, _ = bb._voxel_num_host

        def time_us(fn, reps=20):
            fn(); torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            evs[0].record()
            for r in range(reps):
                fn()
                evs[r + 1].record()
            torch.cuda.synchronize()
            return sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(reps))[reps // 2] * 1e3
        outb = torch.empty(Bn, Y, X, Cvp, device=dev)

        def time_graph_us(fn, reps=20):
            """`reps` launches captured into one hipGraph and replayed: the stream holds them back to back, so the time is the
            GPU's (kernel + the gap between graph nodes), not the cadence of the Python loop -- at ~25 us per launch the
            eager loop above is host-bound on slower hosts.  Falls back to the eager loop if the capture fails."""
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    fn()
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(reps):
                            fn()
                    g.replay()
                    side.synchronize()
                    ts = []
                    for _ in range(5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(side)
                        g.replay()
                        e1.record(side)
                        side.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
                torch.cuda.current_stream(dev).wait_stream(side)
                return sorted(ts)[len(ts) // 2], "hipGraph of 20 launches, HIP events around the replay, median of 5 replays"
            except Exception:
                torch.cuda.synchronize()
                return time_us(fn, reps), "HIP events on the launch stream, median of 20 eager launches"
        pool_us, pool_method = time_graph_us(lambda: plan.pool(feats, out=outb))
        # the form the timed model runs (one camera per sample): no lifted tensor, rows = prob * context inside the same gather
        lift_splat_us = None
        Dh = int(bb.height_channels)
        if getattr(bb, 'fuse_lift_splat', False) and Np % Dh == 0 and int(imgs.shape[2]) == 1:
            Pp = Np // Dh
            prob_t = torch.rand(Bn, Dh, Pp, device=dev)
            ctx_t = torch.randn(Bn, Pp, Cvp, device=dev)
            lift_splat_us, _ = time_graph_us(lambda: plan.lift_splat(prob_t, ctx_t, out=outb))
            del prob_t, ctx_t
        flat = geom.view(Bn, -1, 3)
        build_us = clean_us = None
        if not args.no_plan_timing:
            build_us = time_us(lambda: VoxelPlan(flat, (X, Y, 1), cached=False), reps=10)
            clean_us = time_us(lambda: plan.rebuild(flat), reps=10)
        alg = 12.0 * Bn * Np + 4.0 * Bn * Np * Cvp + 4.0 * Bn * Y * X * Cvp           # geom + feats + output
        kname = {1: "vp_gather_fast_kernel", 2: "vp_gather_vox_kernel"}
        _Lk = __import__("sgv3d_amd._lib", fromlist=["load"]).load()
        k_op = kname.get(_Lk.sgv3d_voxel_pooling_kernel_for(Bn, Np, Cvp, X, Y, 0), "vp_gather3_kernel")
        k_fused = kname.get(_Lk.sgv3d_voxel_pooling_kernel_for(Bn, Np, Cvp, X, Y, 1), "vp_gather3_kernel")
        vtraffic, vsrc = load_vp_traffic(family=k_op)
        # the fused lift-splat launch against ITS algorithmic bytes ("Fused lift-splat"): probabilities +
        # context rows + one linearised index per point + the output written once
        fused_rec = None
        if lift_splat_us:
            alg_f = 4.0 * Bn * Np + 4.0 * Bn * (Np // Dh) * Cvp + 4.0 * Bn * Np + 4.0 * Bn * Y * X * Cvp
            ftraffic, fsrc = load_vp_traffic(fused=True, family=k_fused)
            fused_rec = {"kernel": k_fused + "<.., FUSED> (sgv3d_lift_splat_planned: what the timed model launches)",
                         "bytes": alg_f, "us": lift_splat_us, "achieved": alg_f / lift_splat_us / 1e3, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": alg_f / lift_splat_us / 1e3 / HBM_PEAK_GBPS, "traffic": ftraffic,
                         "traffic_source": fsrc,
                         "note": "not HBM bound: the context map is L2-resident and every point's row is re-read from there; with no row "
                                 "loads and no stores at all the launch still takes 11.5 us at cfg-2 (issue + dependent-load chains, "
                                 "DESIGN 3.2)"}
        # level 1 of INTEGRATION.md: the symbol the reference's own voxel_pooling.py reaches through voxel_pooling_ext
        # (sgv3d_voxel_pooling_forward: device-side compare of geom_xyz + pos_memo, gather accumulating into the caller-zeroed
        # output from the library-owned plan, gated scatter fallback) -- timed on this run's geometry with pos_memo
        from sgv3d_amd import _lib as _L
        lib_ = _L.load()
        outz = torch.zeros(Bn, Y, X, Cvp, device=dev)
        pmz = torch.full((Bn, Np, 3), -1, dtype=torch.int32, device=dev)
        Zv = int(bb._voxel_num_host[2])
        l1 = lambda: lib_.sgv3d_voxel_pooling_forward(Bn, Np, Cvp, X, Y, Zv, flat.data_ptr(), feats.data_ptr(), outz.data_ptr(),
                                                      pmz.data_ptr(), _L.stream_handle(dev))
        level1_us = time_us(l1) if Cvp % 4 == 0 and 24 <= Cvp <= 256 else None
        # ... the same call recorded into a hipGraph (a harness that captures its forward): zero fill (a kernel) + the entry,
        # 20 of them per graph; a capture records the rebuild as ONE gated launch in line (vp_plan_build_one_kernel, csrc/voxel_pooling.hip)
        level1_graph_us = None
        if level1_us is not None:
            def l1_step():
                outz.mul_(0.0)
                l1()
            level1_graph_us, _ = time_graph_us(l1_step)
            zero_us, _ = time_graph_us(lambda: outz.mul_(0.0))
            level1_graph_us = max(level1_graph_us - zero_us, 0.0)       # (the caller's zero fill is not the entry's)
        del outz, pmz
        # level 2 of INTEGRATION.md: the Python operator with the reference's signature, as a caller that swaps only the import
        # uses it -- a fresh output tensor per call, geom_xyz handed in every time (no VoxelPlan object in the caller's hands)
        from sgv3d_amd.ops.voxel_pooling import voxel_pooling as vp_op
        with torch.no_grad():
            op_us = time_us(lambda: vp_op(flat, feats, (X, Y, Zv)))
            op_graph_us, _ = time_graph_us(lambda: vp_op(flat, feats, (X, Y, Zv)))
        roofline_hbm = {
            "bound": "hbm", "kernel": k_op + " (sgv3d_voxel_pooling_forward_planned: one launch, no fix-up pass)",
            "bytes": alg, "us": pool_us, "achieved": alg / pool_us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": alg / pool_us / 1e3 / HBM_PEAK_GBPS, "traffic": vtraffic, "traffic_source": vsrc,
            # the contract's bytes count all N rows; 26 % of the cfg-2 rows fall outside the grid and are never read: the
            # counter traffic over the same launch time is what the memory system actually moved
            "frac_of_counter_traffic": (vtraffic / pool_us / 1e3 / HBM_PEAK_GBPS) if vtraffic else None,
            "plan_build_us": build_us, "plan_check_us": clean_us,
            "frac_including_plan": alg / (pool_us + build_us) / 1e3 / HBM_PEAK_GBPS if build_us else None,
            "frac_including_check": alg / (pool_us + clean_us) / 1e3 / HBM_PEAK_GBPS if clean_us else None,
            "plan_builds_in_timed_region": calib["plan_builds_in_timed_region"],
            "model_path_lift_splat_us": lift_splat_us, "fused_lift_splat": fused_rec,
            "python_op_us": op_us, "frac_python_op": alg / op_us / 1e3 / HBM_PEAK_GBPS,
            "level1_ext_us": level1_us,
            "frac_level1_ext": alg / level1_us / 1e3 / HBM_PEAK_GBPS if level1_us else None,
            "level1_ext_in_graph_us": level1_graph_us,
            "frac_level1_ext_in_graph": alg / level1_graph_us / 1e3 / HBM_PEAK_GBPS if level1_graph_us else None,
            "python_op_in_graph_us": op_graph_us, "frac_python_op_in_graph": alg / op_graph_us / 1e3 / HBM_PEAK_GBPS,
            "note": "the plan depends only on the calibration: built once per calibration outside the captured forward "
                    "(frac_including_plan = if it were rebuilt on every frame, as the reference-style operator call with "
                    "ever-changing geom_xyz would; frac_including_check = operator call with an unchanged geom_xyz: "
                    "device-side compare + empty build launches; frac_level1_ext = the reference wrapper's own call into "
                    "voxel_pooling_ext, which keeps a plan per stream inside the library; frac_python_op = the Python operator "
                    "voxel_pooling(geom_xyz, feats, voxel_num) called eagerly, output allocation and zero fill included; model_path_lift_splat_us = what the timed "
                    "model launches instead of lift + this operator: the same gather forming its rows as prob * context, so the "
                    "[B,N,C] lifted tensor -- most of the operator's bytes -- is neither written nor read)",
            "method": pool_method + ", N(0,1) features on this run's geometry",
        }
        del feats, outb

    # ---- CPU baseline (oracle port), rank 0 at N=1 only -------------------------------------------
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        from oracle import torch_model as TM, voxel_pooling_ref as VPR
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        cimgs, cmats = imgs[:1].cpu(), {k: v[:1].cpu() for k, v in mats.items()}
        cores = torch.get_num_threads()
        # BASELINE.md 3 asks warm-up 3 / median of 10; a cfg-2 frame takes ~5 s on these host cores, so the sample is
        # bounded to 1 warm-up + up to 3 timed frames (<= ~25 s), median reported
        keep = {}
        t0 = time.perf_counter()
        ref = TM.bevheight_forward(sd, bc, hc, cimgs, cmats, keep)          # warm-up frame; also the parity reference
        warm = time.perf_counter() - t0
        times = []
        while len(times) < 3 and sum(times) + warm < 22.0 and not args.sub:      # (child runs: the one frame parity needs)
            t1 = time.perf_counter()
            TM.bevheight_forward(sd, bc, hc, cimgs, cmats)
            times.append(time.perf_counter() - t1)
        med = sorted(times)[len(times) // 2] if times else warm
        # operator-level baseline: the C restatement of voxel_pooling_forward_kernel on this frame's geometry
        g = np.ascontiguousarray(keep['geom_xyz'].reshape(1, -1, 3))
        Cc = int(keep['bev'].shape[1])
        f = np.random.default_rng(0).standard_normal((1, g.shape[1], Cc)).astype(np.float32)
        Xc, Yc, Zc = (int(v) for v in sd['backbone.voxel_num'])
        ob = np.zeros((1, Yc, Xc, Cc), np.float32)

        def cpu_med(threads, reps=5):
            VPR.forward_nhwc_inplace(g, f, ob, Xc, Yc, Zc, threads)
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                VPR.forward_nhwc_inplace(g, f, ob, Xc, Yc, Zc, threads)
                ts.append(time.perf_counter() - t)
            return sorted(ts)[reps // 2] * 1e3
        # the OpenMP restatement partitions the points once (oracle/voxel_pooling_ref.c) and uses every core this process
        # may run on; the best of {all, half, quarter} of them is reported with its thread count
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        omp_ms, ncpu = min((cpu_med(t), t) for t in sorted({max(1, avail), max(1, avail // 2), max(1, avail // 4)}))
        cpu_baseline = {"value": 1.0 / med, "unit": "frames/s", "cores": cores, "kind": "port",
                        "sample": f"1 warm-up + {len(times)} timed full {args.config} frame(s) (batch 1) through "
                                  f"oracle/torch_model.py (torch-CPU fp32 eager + numpy geometry + C voxel pooling), "
                                  f"median {med:.2f} s per frame, {warm + sum(times):.1f} s in total",
                        "voxel_pooling_c_1thread_ms": cpu_med(1), "voxel_pooling_c_openmp_ms": omp_ms,
                        "voxel_pooling_host_cores_available": avail,
                        "voxel_pooling_openmp_threads": ncpu,
                        "voxel_pooling_sample": f"oracle/voxel_pooling_ref.c on this frame's geometry, N={g.shape[1]}, "
                                                f"C={Cc}, warm-up 1, median of 5"}
        # ---- parity of the timed model: same frame, same weights, GPU vs oracle ---------------------
        with torch.no_grad():
            got = model(imgs[:1], {k: v[:1] for k, v in mats.items()})
            ggeom, _ = model.backbone.calibration({k: v[:1] for k, v in mats.items()}, 0)
        torch.cuda.synchronize()
        worst, worst_name, nmaps, scale = 0.0, None, 0, 0.0
        for t in range(len(ref)):
            for k, v in ref[t][0].items():
                e = float((got[t][0][k].float().cpu() - v).abs().max())
                scale = max(scale, float(v.abs().max()))
                nmaps += 1
                if e > worst:
                    worst, worst_name = e, f"task{t}.{k}"
        # fp32: the literal 1e-3 of north_star; bf16 mode: the full-size tests' bar, 2e-2 of the output scale
        # (tests/test_fullsize_gpu.py BF16_TOL; measured 6e-3..7e-3) -- not an absolute figure a 5x regression would pass
        tol = 1e-3 if args.dtype in ("f32", "f32x3", "f32x3auto") else 2e-2 * max(1.0, scale)
        # yardstick: the same forward in float64 (torch on the GPU, a checker): how far is EACH float32 execution from
        # exact arithmetic?  Two float32 implementations differ by their summed rounding noise; the HIP path has to be
        # as close to the float64 result as the reference-style torch-CPU float32 execution is.
        ref64 = TM.bevheight_forward_highprec(sd, bc, hc, cimgs, cmats, device=dev)
        e_hip = max(float((got[t][0][k].double() - ref64[t][0][k]).abs().max()) for t in range(len(ref)) for k in ref[t][0])
        e_cpu = max(float((ref[t][0][k].double() - ref64[t][0][k].cpu()).abs().max()) for t in range(len(ref)) for k in ref[t][0])
        del ref64
        parity = {"max_abs_err": worst, "worst_map": worst_name, "n_maps": nmaps, "max_abs_ref": scale,
                  "voxel_indices_equal": bool(np.array_equal(ggeom.cpu().numpy(), keep['geom_xyz'])),
                  "tolerance": tol, "reference": "oracle/torch_model.py::bevheight_forward on the same frame and weights",
                  "float64_yardstick": {"hip_max_abs_err": e_hip, "oracle_fp32_max_abs_err": e_cpu,
                                        "what": "max |x - float64 forward| over the 36 maps for x = HIP output / torch-CPU fp32 oracle"},
                  "ok": bool(worst <= tol)}
        parity["ok"] = parity["ok"] and parity["voxel_indices_equal"]
        parity["rel_err"] = worst / max(scale, 1e-30)
        # The figures above hold for THESE weights (last BatchNorm of every residual block at gamma x 0.3, activations
        # O(1-10)).  With gamma ~ 1 (untrained statistics, activations ~1e2) two float32 executions of the network differ
        # by more than 1e-3 ABSOLUTE while their RELATIVE distance stays ~1e-5: reported here so that nobody reads the
        # absolute figure as a property of arbitrary checkpoints ("float32 noise and the 1e-3 bar").
        if args.dtype == "f32" and not args.sub:
            torch.manual_seed(0)
            m1 = BEVHeight(bc, hc).eval()
            S.randomize_norm_stats_(m1, 0)                       # residual_gamma = 1
            sd1 = {k: v.detach().cpu() for k, v in m1.state_dict().items()}
            ref1 = TM.bevheight_forward(sd1, bc, hc, cimgs, cmats)
            m1 = m1.to(dev)
            with torch.no_grad():
                got1 = m1(imgs[:1], {k: v[:1] for k, v in mats.items()})
            torch.cuda.synchronize()
            e1 = max(float((got1[t][0][k].float().cpu() - ref1[t][0][k]).abs().max()) for t in range(len(ref1)) for k in ref1[t][0])
            s1 = max(float(ref1[t][0][k].abs().max()) for t in range(len(ref1)) for k in ref1[t][0])
            parity["gamma1_weights"] = {"max_abs_err": e1, "max_abs_ref": s1, "rel_err": e1 / max(s1, 1e-30),
                                        "what": "same check with the residual-block BatchNorm gamma left at ~1 (activations ~1e2): "
                                                "not gated, float32 noise of both executions (DESIGN.md section 4)"}
            del m1, got1, ref1, sd1

    # ---- one frame in flight with the kernels chosen FOR one frame in flight ---------------------------------------
    # `single` above replays one frame at a time with the tiles / Winograd variants that were picked for `nstreams` frames in
    # flight.  A caller who runs one frame at a time builds its pipeline with one slot and gets its own choices: measured here,
    # last (it drops every packed weight and measurement of the model), so both uses of the library are on the record.
    single_own = None
    if single is not None and hip_ops.TUNE_STREAMS > 1 and hip_ops.AUTOTUNE:     # (a tune cache was saved right after the warm-up)
        saved_streams = hip_ops.TUNE_STREAMS
        hip_ops.TUNE_STREAMS = 1             # (layer signatures carry the load they were measured under: "|ts1")
        model.refresh()                      # drops the packed weights and with them every layer's cached choice
        one = FramePipeline(model, imgs, mats, slots=1, use_graph=use_graph)     # its first forward re-measures, alone
        for _ in range(args.warmup):
            one.submit(imgs, mats)
        t1 = group.timed(lambda: one.submit(imgs, mats), args.steps)
        single_own = {"value": B * args.steps / t1, "ms_per_step": t1 / args.steps * 1e3}
        del one
        hip_ops.TUNE_STREAMS = saved_streams
        hip_ops.save_tune_db()               # (SGV3D_TUNE_CACHE only) now also holds the one-frame-in-flight choices

    # ---- the reference harness's eval_step AS IT IS WRITTEN (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:
    # 242-258; sgv3d_amd/harness.py): per step fresh `.cuda()` calibration tensors from host tensors, ONE `model(imgs, mats)`
    # on the caller's stream, get_bboxes, `.cpu().numpy()` of boxes / scores / labels per sample (host syncs).  No
    # FramePipeline: BEVHeight.forward answers from its own per-signature hipGraph (models/bev_height.py).  One frame in
    # flight by construction (the harness waits for every frame's boxes), images resident in HBM like every other figure here.
    # Runs last, with the per-layer choices of ONE frame in flight (TUNE_STREAMS = 1, what a process that never builds a
    # FramePipeline has), i.e. the same kernels as one_frame_in_flight_own_tiles_value.
    harness_rec = None
    if rank == 0 and world == 1 and not args.sub and not stub and not args.no_harness:
        from sgv3d_amd import harness as H
        saved_streams = hip_ops.TUNE_STREAMS
        if hip_ops.TUNE_STREAMS != 1:
            hip_ops.TUNE_STREAMS = 1
            model.refresh()
        host_mats = {k: v.cpu() for k, v in mats.items()}
        hstep = lambda: H.eval_step(model, H.make_batch(imgs, host_mats))
        with torch.no_grad():
            default_mode = model.graph_forward                  # "auto": the model keeps the replay where it measures faster
            # (warm-up: call 1 runs eagerly and measures, call 2 captures the model's own graph and times replay against eager,
            # dropping the graph's ~1 GB activation pool if eager wins; a few more settle the caching allocator after that)
            for _ in range(max(8, args.warmup)):
                res = hstep()
            th = group.timed(hstep, args.steps)
            n_h = max(args.steps, int(1.2 / max(th / args.steps, 1e-4)) + 1)
            th_long = group.timed(hstep, n_h)
            chosen = next((e for e in model._graphs.values() if len(e) > 2), None)
            forced = {}
            for mode in (True, False):                          # and both forms forced, for the record
                model.graph_forward = mode
                model._graphs = {}
                for _ in range(3):
                    hstep()
                forced[mode] = group.timed(hstep, args.steps)
            graph_entry = next((e[1] for e in model._graphs.values() if e[1]), None)
            model.graph_forward = default_mode
            model._graphs = {}
        hip_ops.TUNE_STREAMS = saved_streams
        harness_rec = {"value": B * args.steps / th, "ms_per_step": th / args.steps * 1e3,
                       "long_run_value": B * n_h / th_long, "long_run_steps": n_h,
                       "forward_mode": str(default_mode), "auto_choice": chosen[2] if chosen else None,
                       "graph_forward_value": B * args.steps / forced[True], "graph_forward_ms_per_step": forced[True] / args.steps * 1e3,
                       "eager_forward_value": B * args.steps / forced[False], "eager_forward_ms_per_step": forced[False] / args.steps * 1e3,
                       "boxes_last_frame": int(res[0][0].shape[0]),
                       "what": "sgv3d_amd/harness.py::eval_step = the reference Lightning module's eval_step (exps/...:242-258): fresh "
                               ".cuda() calibration tensors, model(imgs, mats), get_bboxes, .cpu().numpy() x3 per sample; one frame in "
                               "flight, host sync per step; forward = BEVHeight's default (graph_forward 'auto': its own hipGraph replay where "
                               "that measures faster than the eager call on this host, auto_choice); graph_forward_value / "
                               "eager_forward_value: SGV3D_GRAPH_FORWARD=1 / 0"}

    # ---- the same eval_step at batch 8: the reference's documented evaluation geometry is `-e -b 8 --gpus 8`
    # (docs/run_and_eval.md:5-10) -- eight frames per step and GPU.  Same unchanged harness loop, same model object.
    harness_b8 = None
    if harness_rec is not None and args.config == "cfg2" and B == 1 and not args.no_harness_b8:
        try:
            from sgv3d_amd import harness as H
            B8 = 8
            imgs8 = S.make_images(B8, bc['final_dim'], device=dev, seed=1)
            host8 = {k: v.cpu() for k, v in S.make_mats(B8, device=dev).items()}
            saved_streams = hip_ops.TUNE_STREAMS
            hip_ops.TUNE_STREAMS = 1
            try:
                m0 = hip_ops.TUNE_STATS["measured"]
                hstep8 = lambda: H.eval_step(model, H.make_batch(imgs8, host8))
                with torch.no_grad():
                    for _ in range(4):
                        res8 = hstep8()
                    n8 = max(5, args.steps // 2)
                    t8 = group.timed(hstep8, n8)
            finally:
                # whatever happened: the later sections run under the switches (and without the graphs) they would have had
                hip_ops.TUNE_STREAMS = saved_streams
                model._graphs = {}
            harness_b8 = {"value": B8 * n8 / t8, "unit": "frames/s", "ms_per_step": t8 / n8 * 1e3, "batch": B8, "steps": n8,
                          "layers_measured_for_batch_8": hip_ops.TUNE_STATS["measured"] - m0,
                          "boxes_per_frame": [int(r[0].shape[0]) for r in res8],
                          "what": "harness eval_step (exps/...:242-258) with 8 frames per step, the batch size docs/run_and_eval.md:5-10 "
                                  "evaluates with; one step in flight, host sync per step"}
            del imgs8, res8
            hip_ops.save_tune_db()           # (SGV3D_TUNE_CACHE only) ... and the batch-8 signatures
        except Exception as e:          # reported, not fatal: the judged figure is batch 1
            harness_b8 = {"error": repr(e)[:300]}

    # ---- BASELINE configs[2] / [4] in their own dtype, as compact records (child runs of this script) ---------------
    other_configs = None
    if rank == 0 and world == 1 and args.config == "cfg2" and args.dtype == "f32" and not args.sub and not args.no_other_configs \
            and not args.no_cpu_baseline:
        other_configs = run_other_configs(args)

    # ---- the training step and the all-f32-MFMA value, as child runs (one GPU process at a time) ------------------------------
    train_steps = native_f32 = None
    main_line = rank == 0 and world == 1 and args.config == "cfg2" and args.dtype == "f32" and not args.sub and not stub
    if main_line and not args.no_train_step and not args.no_cpu_baseline:
        train_steps = run_train_steps()
    x3_layers = None
    if not stub:
        x3_layers = sum(1 for k, v in hip_ops.TUNE_DB.items() if v[0] in hip_ops.WINO4_X3_TILES + hip_ops.PW_X3_TILES and k.endswith(f"|ts{hip_ops.TUNE_STREAMS}")
                        and f"|{B}x" in k)
    if main_line and (hip_ops.WINO4_X3 or hip_ops.PW_X3) and not args.no_native_f32:
        native_f32 = run_native_f32(args)

    if rank == 0:
        products = ("f32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)" if not ((hip_ops.WINO4_X3 or hip_ops.PW_X3) and args.dtype == "f32") else
                    "f32-accurate: every product of the three-launch F(4x4) layers' position GEMMs (csrc/gemm_x3_grouped.hip) and of the 1x1 "
                    "layers (csrc/conv_pw_x3.hip) is the f32 sum of six bf16 x bf16 partial products of operands split exactly into three "
                    "bf16 terms (error of a product: one f32 rounding, f32 accumulation) where the per-layer measurement picked that form; "
                    "every other product on the f32 MFMA (native_f32_value: all of them)")
        line = {
            "metric": "camera frames/sec at 864x1536->BEV",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "stub" if stub else "synthetic",
            "config": {"workload": workload, "products": products, "layers_on_f32x3_tiles_in_tune_db": x3_layers,
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"replicas x{world}",
                       "hip_graph": bool(use_graph), "frames_in_flight": nstreams,
                       "one_frame_in_flight_value": single["value"] if single else None,
                       "one_frame_in_flight_ms_per_step": single["ms_per_step"] if single else None,
                       "fuse_lift_splat": not args.no_fuse_lift_splat,
                       "voxel_pooling_mode": "planned, plan cached per calibration",
                       "calibration_cache": calib,
                       "weights": "random-init (seed 0), BN statistics / affine perturbed, last BN of every residual block scaled by 0.3 "
                                  "(mmdet zero-initialises it) so that activations stay O(1-10)",
                       "world_size": group.dist.get_world_size() if group.dist is not None else 1,
                       "backend": (group.backend or "none") + (" (RCCL)" if group.backend == "nccl" else ""),
                       "per_rank": per_rank},
            "one_frame_in_flight_value": single["value"] if single else None,
            "one_frame_in_flight_ms_per_step": single["ms_per_step"] if single else None,
            "one_frame_in_flight_own_tiles_value": single_own["value"] if single_own else None,
            "one_frame_in_flight_own_tiles_ms_per_step": single_own["ms_per_step"] if single_own else None,
            "long_run_value": long_run["value"] if long_run else None, "long_run": long_run,
            "harness_eval_step_value": harness_rec["value"] if harness_rec else None, "harness_eval_step": harness_rec,
            "harness_eval_step_b8": harness_b8,
            "native_f32_value": native_f32["value"] if native_f32 and "value" in native_f32 else None, "native_f32": native_f32,
            "train_step": train_steps,
            "cached_calibration_value": cached["value"] if cached else None,
            "cached_calibration": cached,
            "changing_calibration_every_frame_value": changing["value"] if changing else None,
            "changing_calibration_every_frame": changing,
            "roofline": roofline, "roofline_hbm": roofline_hbm, "cpu_baseline": cpu_baseline, "parity": parity,
            "other_configs": other_configs,
        }
    # rank 0 has been busy alone since the last collective (roofline pass, CPU baseline): every rank waits here, so that
    # no rank tears RCCL down while another one is still inside the job
    group.barrier()
    group.close()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)        # C stdio buffers (the RCCL banner) go to stderr, not after our line
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        # a child that crashed, timed out, printed no JSON or was skipped has no exit_code / parity: that is a failure too
        bad = [o for o in (other_configs or [])
               if "error" in o or "skipped" in o or o.get("exit_code") != 0 or (o.get("parity") or {}).get("ok") is not True]
        if bad:
            print(f"[bench] PARITY FAILURE in other_configs: {bad}", file=sys.stderr)
            sys.exit(3)
        if parity is not None and not parity["ok"]:
            print(f"[bench] PARITY FAILURE: max |hip - oracle| = {parity['max_abs_err']:.3e} on {parity['worst_map']} "
                  f"(tolerance {parity['tolerance']}), voxel indices equal: {parity['voxel_indices_equal']}", file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
