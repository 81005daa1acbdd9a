"""N>1 host path on CPU: world_size-2 gloo run of the frame sharding + keypoint-count gather used by bench.py."""
import importlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    d = importlib.import_module("orb-slam3_amd.dist")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = d.shard_frames(n_frames, rank, world)
    counts = [1000 + 7 * f for f in mine]                 # stand-in for the per-frame keypoint counts of this rank's GPU
    gathered = d.gather_counts(counts)
    full = d.reassemble(gathered, n_frames)
    tmax = torch.tensor([float(rank + 1)])
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)           # bench.py's max-over-ranks timing
    q.put((rank, full.tolist(), float(tmax.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 7, 1])
def test_sharding_and_count_gather_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sock:                          # a port that is free right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, tmax in res:
        assert full == [1000 + 7 * f for f in range(n_frames)]
        assert tmax == 2.0


def test_shard_is_a_partition():
    d = importlib.import_module("orb-slam3_amd.dist")
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 4, 8):
            allf = sorted(sum((d.shard_frames(n, r, w) for r in range(w)), []))
            assert allf == list(range(n))


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` outside torchrun spawns 2 fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*),
    which rendezvous (gloo here), take the max-over-ranks time and gather the per-frame counts; rank 0 prints the one JSON
    line with n_gpus = 2.  ORB_BENCH_STUB=1 swaps the GPU workload for a sleep, so this runs without a GPU."""
    import json
    import subprocess
    env = dict(os.environ, ORB_BENCH_STUB="1", ORB_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["frames_per_step_per_gpu"] == 5
    assert d["config"]["keypoints_last_batch"] == 5 * 1000 + 5 * 1001        # both ranks' counts arrived
    assert abs(d["value"] - 2 * 5 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]   # whole-job frames / max-over-ranks time


def test_bench_launcher_reports_a_failed_rank():
    import subprocess
    env = dict(os.environ, ORB_BENCH_STUB="1", ORB_BENCH_BACKEND="gloo", ORB_BENCH_STUB_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_bench_launcher_eight_ranks_and_a_straggler():
    """The shape of the driver's 8-GPU run, rehearsed on CPU: 8 rank processes, one JSON line with 8 per-rank step times in which
    the slow rank stands out and sets the whole-job value (max over ranks)."""
    import json
    import subprocess
    env = dict(os.environ, ORB_BENCH_STUB="1", ORB_BENCH_BACKEND="gloo", ORB_BENCH_STUB_SLOW_RANK="5")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "1", "--batch", "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and len(d["per_rank_ms_per_step"]) == 8
    pr = d["per_rank_ms_per_step"]
    assert max(range(8), key=lambda r: pr[r]) == 5 and pr[5] > 15.0 and max(pr[r] for r in range(8) if r != 5) < 10.0
    assert d["ms_per_step"] >= pr[5] - 1e-6                        # the whole job runs at the straggler's pace
    assert d["config"]["keypoints_last_batch"] == sum(3 * (1000 + r) for r in range(8))
