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


def _bench(args, env_extra=None, timeout=180):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (gloo rehearsal of the plumbing: no
    device work), shards the streams over them and prints ONE line with n_gpus = 2 and the whole-job totals."""
    rc, line, err = _bench(["--gpus", "2", "--rehearse", "--steps", "4", "--streams", "5"])
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["rehearsal"] is True
    assert line["streams_total"] == 10                 # 5 per rank, SUM over ranks
    assert line["ms_per_step"] >= 2.0                  # MAX over ranks: rank 1 sleeps 2 ms per step
    assert line["id_mismatch"] == 0


def test_bench_launcher_reports_sentinel_mismatch_and_rank_failure():
    rc, line, err = _bench(["--gpus", "3", "--rehearse", "--steps", "2"], {"MSKF_REHEARSE_CORRUPT_RANK": "2"})
    assert rc == 0 and line["n_gpus"] == 3 and line["id_mismatch"] == 1
    # a world size that contradicts --gpus is refused (n_gpus in the line is never a guess)
    rc, line, err = _bench(["--gpus", "4", "--rehearse"], {"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc == 2 and line is None and "WORLD_SIZE" in err
