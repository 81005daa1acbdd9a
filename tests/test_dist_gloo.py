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
