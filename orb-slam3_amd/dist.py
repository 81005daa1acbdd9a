"""Frame-parallel multi-GPU plumbing: one process per GPU, frames (or stereo pairs) are independent
units, so the data path needs NO collective; the only exchange is the gather of per-frame keypoint counts (and,
for a consumer on rank 0, of the padded keypoint/descriptor blocks).  Works with any torch.distributed backend:
"nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Frame f of a batch goes to GPU f mod G; returns this rank's frame indices."""
    return list(range(rank, n_frames, world))


def gather_counts(local_counts, device=None):
    """all_gather of int32 keypoint counts; ranks may own different numbers of frames (padded with -1).
    Returns a list (per rank) of 1-D int32 tensors on the caller's device."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.as_tensor(local_counts, dtype=torch.int32, device=device)
    if world == 1:
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    padded = torch.full((m,), -1, dtype=torch.int32, device=t.device)
    padded[:t.numel()] = t
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return [o[:int(s.item())] for o, s in zip(out, sizes)]


def reassemble(per_rank_counts, n_frames):
    """Inverse of shard_frames for the gathered counts: counts in original frame order."""
    world = len(per_rank_counts)
    res = torch.full((n_frames,), -1, dtype=torch.int32)
    for r, c in enumerate(per_rank_counts):
        idx = shard_frames(n_frames, r, world)
        res[idx] = c.cpu()
    return res
