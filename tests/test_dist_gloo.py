"""N > 1 path on CPU: world_size-2 gloo processes exercise the only cross-rank logic of the benchmark
(barrier + MAX/SUM reduction; stream sharding) — the data path itself has no collective."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from msckf_stereo_c_amd.dist_util import aggregate_throughput, shard_streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(10, world, rank)
    dist.barrier()
    elapsed, frames = aggregate_throughput(1.0 + rank, len(mine) * 7, world)
    q.put((rank, mine, elapsed, frames))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, f0), (r1, s1, e1, f1) = res
    assert sorted(s0 + s1) == list(range(10)) and not set(s0) & set(s1)      # every stream on exactly one rank
    assert e0 == e1 == 2.0                                                     # MAX over ranks
    assert f0 == f1 == 70.0                                                    # SUM over ranks
