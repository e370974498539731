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
