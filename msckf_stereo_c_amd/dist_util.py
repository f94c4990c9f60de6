"""Multi-GPU aggregation for the sharded-stream benchmark.

VIO streams are independent units: stream s lives on exactly one rank, nothing of the data path crosses
ranks.  The only collectives are the barrier around the timed region and this reduction of two scalars
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def aggregate_throughput(elapsed_s, frames_done, world, device="cpu"):
    """MAX of the per-rank elapsed time, SUM of the per-rank processed stereo frames."""
    if world <= 1 or not dist.is_initialized():
        return float(elapsed_s), float(frames_done)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(frames_done)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())


def shard_streams(n_streams_total, world, rank):
    """Stream s -> rank s mod world; returns the global stream ids of this rank."""
    return [s for s in range(n_streams_total) if s % world == rank]
