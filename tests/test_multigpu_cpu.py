"""CPU: the N>1 launch path of bench.py / the replica sharding logic, world_size 2 on gloo.

Inference shards frames over GPUs as independent replicas: no collective on the data
path, only a barrier + MAX-reduce of the elapsed time around the timed region.  This test runs that
exact protocol (sgv3d_amd.replicas) with two gloo processes and a stand-in step function."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from sgv3d_amd.replicas import ReplicaGroup, shard_frames

    g = ReplicaGroup(backend="gloo")
    assert g.world == 2 and g.rank == int(os.environ["RANK"])
    # frames shard without overlap and cover the global batch
    mine = shard_frames(list(range(7)), g.rank, g.world)
    gathered = g.all_gather_object(mine)
    assert sorted(sum(gathered, [])) == list(range(7)), gathered
    assert all(len(x) in (3, 4) for x in gathered)
    # timed region protocol: barrier, K steps, barrier, MAX over ranks
    def step():
        time.sleep(0.01 * (g.rank + 1))       # rank 1 is slower
    elapsed = g.timed(step, steps=5)
    assert elapsed >= 0.05 * 2 * 0.9, elapsed  # the MAX (rank 1's 0.1 s), same on both ranks
    agree = g.all_gather_object(round(elapsed, 6))
    assert agree[0] == agree[1]
    value = g.aggregate_throughput(units_per_rank_per_step=3, steps=5, elapsed=elapsed)
    assert abs(value - 2 * 3 * 5 / elapsed) < 1e-9
    g.close()
    if g.rank == 0:
        print(json.dumps({"ok": True, "elapsed": elapsed}))
""") % ROOT


def test_two_process_gloo_replicas(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b'"ok": true' in outs[0][0]


def test_shard_frames_single_process():
    from sgv3d_amd.replicas import shard_frames
    assert shard_frames(list(range(5)), 0, 1) == [0, 1, 2, 3, 4]
    parts = [shard_frames(list(range(10)), r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10)) and max(map(len, parts)) - min(map(len, parts)) <= 1


TRAIN_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from sgv3d_amd.train_step import FlatParams, DataParallelAdamW
    from sgv3d_amd._lib import SGV3DError

    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.manual_seed(7 * rank)                              # DIFFERENT initial weights per rank ...
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 5, 1))
    mine = [p.detach().clone() for p in net.parameters()]
    # a frozen stem beside the trainable net (frozen_stages=0 of the reference's image backbone): in no bucket, on either rank
    stem = torch.nn.Conv2d(3, 3, 1)
    for p in stem.parameters():
        p.requires_grad = False
    stem_before = [p.detach().clone() for p in stem.parameters()]
    opt = DataParallelAdamW(list(stem.parameters()) + list(net.parameters()), lr=1e-3, bucket_bytes=1024, max_grad_norm=5.0)     # tiny buckets: several collectives
    assert len(opt.flat.buckets) >= 2
    bucketed = {id(p) for _, _, entries in opt.flat.buckets for p, _, _ in entries}
    assert not any(id(p) in bucketed for p in stem.parameters()) and all(id(p) in bucketed for p in net.parameters())
    assert all(torch.equal(a, b) for a, b in zip(stem_before, stem.parameters()))     # (not broadcast either: constants of the checkpoint)
    both = [None, None]
    dist.all_gather_object(both, [p.tolist() for p in mine])
    for i, p in enumerate(net.parameters()):                  # ... and the constructor broadcast rank 0's (what DDP does)
        assert torch.equal(p.detach(), torch.tensor(both[0][i])), i
    opt.check_replicas()
    if rank == 1:
        with torch.no_grad():
            next(net.parameters()).add_(1e-3)                 # drift on one rank
    try:
        opt.check_replicas()
        raise AssertionError("drift not detected")
    except SGV3DError as e:
        assert "drifted" in str(e)
    opt.broadcast_parameters()
    opt.check_replicas()
    torch.manual_seed(100 + rank)                            # different data per rank
    x = torch.randn(4, 3, 10, 10)
    opt.zero_grad()
    net(x).square().mean().backward()                        # autograd accumulates into the bucket views
    opt.flat.check_views()
    local = [p.grad.detach().clone() for p in net.parameters()]
    opt.all_reduce_grads()
    for w in opt._pending:
        w.wait()
    gathered = [None, None]
    dist.all_gather_object(gathered, [g.tolist() for g in local])
    for i, p in enumerate(net.parameters()):
        want = torch.tensor(gathered[0][i]) + torch.tensor(gathered[1][i])
        assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), i
    # overlap mode: the collectives are launched from inside backward, bucket by bucket
    opt.overlap_with_backward()
    opt.zero_grad()
    net(x).square().mean().backward()
    assert len(opt._early) == len(opt.flat.buckets)           # every bucket went out during backward
    opt.all_reduce_grads()
    for w in opt._pending:
        w.wait()
    for i, p in enumerate(net.parameters()):
        want = torch.tensor(gathered[0][i]) + torch.tensor(gathered[1][i])
        assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), ("overlap", i)
    assert opt._early == {} and opt._left == [len(e) for _, _, e in opt.flat.buckets]
    try:
        opt.step()
        raise AssertionError("the CPU must not have an optimiser path")
    except SGV3DError as e:
        assert "GPU" in str(e) or "missing" in str(e)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": True}))
""") % ROOT


def test_two_process_gloo_gradient_buckets(tmp_path):
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b'"ok": true' in outs[0][0]


def test_learning_rate_rules():
    from sgv3d_amd.train_step import reference_lr, multistep_lr
    assert abs(reference_lr(4, 8) - 2e-4 / 64 * 32) < 1e-15          # config 4: global batch 32
    assert multistep_lr(1.0, 0) == 1.0 and abs(multistep_lr(1.0, 19) - 0.1) < 1e-12 and abs(multistep_lr(1.0, 23) - 0.01) < 1e-12


def test_overlap_hooks_allow_one_backward_per_step():
    """overlap_with_backward(): a second backward before step() would add local gradients on top of buckets whose
    all-reduce is already in flight -- it raises; zero_grad() / all_reduce_grads() re-arm the counters."""
    import torch
    from sgv3d_amd.train_step import DataParallelAdamW
    from sgv3d_amd._lib import SGV3DError
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    opt = DataParallelAdamW(net.parameters(), lr=1e-3, bucket_bytes=64).overlap_with_backward()
    assert len(opt.flat.buckets) >= 2
    x = torch.randn(4, 6)
    opt.zero_grad()
    net(x).sum().backward()
    assert opt._left == [0] * len(opt.flat.buckets) and opt._early == {}      # no process group: nothing launched
    try:
        net(x).sum().backward()
        raise AssertionError("second backward not detected")
    except SGV3DError as e:
        assert "second backward" in str(e)
    opt.zero_grad()                                                            # re-arms
    net(x).sum().backward()
    opt.all_reduce_grads()
    assert opt._left == [len(e) for _, _, e in opt.flat.buckets]


def test_bench_multi_rank_control_flow_with_stub_model():
    """bench.py's own N > 1 protocol, launched the way the driver launches it (RANK / WORLD_SIZE / MASTER_* in the
    environment, --gpus 2), over gloo with the model replaced by a sleep (SGV3D_BENCH_STUB=1): both ranks finish, rank 0
    prints exactly one JSON line, the elapsed time is the MAX over ranks (rank 1 is made the slower one), every rank's own
    record is gathered, and nobody leaves the process group while rank 0 is still working."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29523", WORLD_SIZE="2", SGV3D_BENCH_STUB="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    lines = [l for l in outs[0][0].decode().splitlines() if l.strip()]
    assert len(lines) == 1 and not outs[1][0].strip()           # ONE line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["data"] == "stub" and rec["scaling"] == "weak"
    assert rec["config"]["world_size"] == 2 and rec["config"]["backend"].startswith("gloo")
    per = rec["config"]["per_rank"]
    assert [r["rank"] for r in per] == [0, 1]
    # rank 1 sleeps 4 ms per step, rank 0 2 ms: the aggregate is priced with the slower rank's time
    assert rec["ms_per_step"] >= 4.0 and per[0]["frames_per_s"] > per[1]["frames_per_s"]
    assert abs(rec["value"] - 2 * 1 * 6 / (rec["ms_per_step"] * 6e-3)) < 1e-6 * rec["value"]


def test_bench_gpus_n_without_launcher_spawns_its_own_ranks():
    """``python bench.py --gpus 2`` with NO rank environment (a user, not torchrun): the script starts its two ranks itself as
    fresh child processes and prints their one line with n_gpus 2 -- it must never run one rank silently and report n_gpus 1.
    Without the stub (a real run) and fewer than N visible GPUs it exits non-zero before any GPU call."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=dict(env, SGV3D_BENCH_STUB="1"), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size"] == 2 and [p["rank"] for p in rec["config"]["per_rank"]] == [0, 1]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                           env=env, capture_output=True, timeout=300)
        assert r.returncode != 0 and not r.stdout.strip()
        assert b"GPU(s) visible" in r.stderr
    # a launcher whose world size disagrees with --gpus is refused as well (used to pass silently for WORLD_SIZE=1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=dict(env, SGV3D_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, timeout=300)
    assert r.returncode != 0 and b"must agree" in r.stderr


def test_bench_world_8_stub_every_rank_reports_its_tune_db():
    """The driver's 8-GPU launch, rehearsed on the CPU: eight gloo ranks of bench.py with the stub model.  One JSON line, n_gpus 8,
    eight per-rank records in rank order, each bound to its own local rank, each with the committed tune DB loaded (so no rank
    would spend its first forward timing candidates under 8-way host contention) and nothing measured."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1"],
                       env=dict(env, SGV3D_BENCH_STUB="1", OMP_NUM_THREADS="1"), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    per = rec["config"]["per_rank"]
    assert rec["n_gpus"] == 8 and rec["config"]["world_size"] == 8 and rec["config"]["global_batch"] == 8
    assert [p["rank"] for p in per] == list(range(8)) and [p["local_rank"] for p in per] == list(range(8))
    assert all(p["tune_db_entries"] > 100 and p["layers_measured_here"] == 0 for p in per)
    # the slowest rank (rank 7 sleeps 16 ms per step) prices the aggregate
    assert rec["ms_per_step"] >= 16.0
    assert abs(rec["value"] - 8 * 4 / (rec["ms_per_step"] * 4e-3)) < 1e-6 * rec["value"]
