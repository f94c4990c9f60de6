#!/usr/bin/env python3
"""bench.py — throughput of the MI355X hot path (stereo KLT front-end + MSCKF update) on synthetic
EuRoC-shaped streams.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

One process per GPU.  With --gpus N > 1 and no WORLD_SIZE in the environment this process only SPAWNS the N ranks
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) before importing torch or touching HIP, relays
rank 0's JSON line and exits non-zero if any rank fails; under torch.distributed.run the ranks are already there.
RCCL is used only for the barrier and the scalar reductions {time MAX, frames SUM, sentinel-hash gather}: the streams
are independent units (stream s -> rank s mod N), no data-path collective.
A "step" is one stereo frame of EVERY stream of the batch: full front-end (pyramids, detector, temporal LK, stereo
LK, gates) + back-end (IMU propagation, augmentation, triangulation, Jacobians, gating, QR, Kalman update, clone
pruning).  Inputs (rendered stereo pairs) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per HIP stream (8 groups x {front-end, filter}); the ROCm default of 4 multiplexes the
# 16 streams and serialises unrelated kernels.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

# SURVEY.md §8(d) algorithmic work per unit
LK_BYTES_PER_TRACK = 4 * (17 * 17 + 16 * 16)      # 4 levels x (17x17 template + 16x16 search footprint) u8
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6                           # FP64 matrix peak
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2                # wave-instructions/s: 1024 SIMD-32, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)


def _profile(name):
    """Newest committed summary of a counter pass (profiles/rNN_<name>.json)."""
    for rnd in ("r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r04_%s.json" % name)


PMC_TRAFFIC_FILE = _profile("pmc_hbm_traffic")
PMC_SQ_FILE = _profile("pmc_sq")
PMC_MFMA_FILE = _profile("pmc_mfma_c2")

PROFILE_NAMES = {"k_pyr_down": "k_pyr_down3", "k_ekf_small": "k_ekf_small_update"}    # timing entry -> kernel name in the counter summaries

COMPRESSION_MODES = {"auto": 0, "gram": 1, "tsqr": 2, "householder": 3}      # mskf_ekf_cfg.compression_mode (include/mskf_types.h)

# BASELINE.json configs: image, clone window, grid rows x cols x min x max, streams per GPU, groups
CONFIGS = {
    "c2": dict(width=752, height=480, clones=30, grid="8x10x5x6", streams=1536, groups=8, loop=60, cpu_frames=150, cpu_all_frames=80,
               name="configs[1] Single MI355X: 752x480 stereo, 30 cam clones, 400 features/frame, 200 Hz IMU"),
    "c3": dict(width=1280, height=720, clones=50, grid="10x20x4x5", streams=768, groups=8, loop=60, cpu_frames=60, cpu_all_frames=30,
               name="configs[2] Single MI355X stress: 1280x720 stereo, 50 cam clones, 1000 features/frame"),
    "c4": dict(width=752, height=480, clones=30, grid="8x10x5x6", streams=8, groups=1, loop=100, cpu_frames=150, cpu_all_frames=80,
               name="configs[3] 8xMI355X: 64 EuRoC-shaped streams sharded 8/GPU"),
    "c5": dict(width=3840, height=2160, clones=60, grid="20x25x4x5", streams=64, groups=4, loop=30, cpu_frames=40, cpu_all_frames=6,
               name="configs[4] 4K stereo streams, 2000 features/frame, 60 cam clones"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json configuration (c2 = the metric's own)")
    ap.add_argument("--streams", type=int, default=None, help="VIO streams per GPU")
    ap.add_argument("--groups", type=int, default=None, help="host thread groups per GPU")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("MSKF_BENCH_HOST_THREADS", "1")), help="host threads per group")
    ap.add_argument("--no-pipeline", action="store_true", help="run front-end and filter of a group in lockstep on one thread")
    ap.add_argument("--unique", type=int, default=0, help="distinct rendered sequences per GPU (streams of a group cycle over them); "
                    "0 = one per stream of a group, so that no two streams of a launch read the same image")
    ap.add_argument("--stagger", type=int, default=-1, help="group g runs g*STAGGER frames ahead, so that the groups replaying the same "
                    "sequences never read the same stereo pair at the same time (-1 = auto: loop // groups + 1, 0 = off)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--clones", type=int, default=None)
    ap.add_argument("--grid", type=str, default=None, help="rows x cols x min x max features per cell")
    ap.add_argument("--prime", type=int, default=None, help="untimed frames before warmup: gravity init + clone window fill")
    ap.add_argument("--loop", type=int, default=None, help="frames per trajectory period")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the 1-core CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--cpu-all-frames", type=int, default=None, help="frames per stream of the all-cores CPU-oracle leg (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("MSKF_BENCH_BACKEND", "nccl"), help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--compression", choices=["householder", "auto", "gram", "tsqr"], default="householder",
                    help="QR compression of the stacked Jacobian (msckf_vio.cpp:795-821): householder = the reference's own rule, Householder QR when "
                         "the stack has more rows than columns and nothing otherwise (default); auto = Gram + regularised Cholesky with the "
                         "Householder TSQR where the device decides it is needed; gram = Gram only; tsqr = Householder always")
    ap.add_argument("--gram-steps", type=int, default=-1,
                    help="steps of the SECOND timed window, run with the Gram + regularised Cholesky compression (--compression auto) instead of the "
                         "reference's Householder QR (value_gram_cholesky in the JSON line); -1 = min(steps, 12) when the first window ran with "
                         "--compression householder, 0 = off")
    ap.add_argument("--host-images", action="store_true", help="stereo pairs stay in host memory: PCIe-inclusive rate (not the headline value)")
    ap.add_argument("--host-render", action="store_true", help="render the synthetic sequences on the host (round 3's way) instead of on the device")
    ap.add_argument("--rehearse", action="store_true", help="CPU rehearsal of the multi-rank plumbing (launcher, rendezvous, reductions, "
                    "JSON line) without any device work: for the gloo tests only, the line it prints says so")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config]
    for k in ("width", "height", "clones", "grid", "streams", "groups", "loop", "cpu_frames", "cpu_all_frames"):
        if getattr(args, k) is None:
            setattr(args, k, preset[k] if k not in ("streams", "groups") else int(os.environ.get("MSKF_BENCH_" + k.upper(), preset[k])))
    if args.gram_steps < 0:
        args.gram_steps = min(args.steps, 12) if (args.compression == "householder" and not args.no_pipeline) else 0
    if args.prime is None:
        # static start, gravity / bias initialisation, clone window full — and one whole period of the looping trajectory, so that
        # every group has met its largest frame (staging buffers grow to their final size) before the timed steps
        args.prime = max(25 + 20 + args.clones, 25 + args.loop + 2)
    return args


# ------------------------------------------------------------------------------------------ rank launcher
def launch_ranks(args, argv):
    """Spawn one rank per GPU (no torch / HIP in this process), relay rank 0's line, fail if any rank fails."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:          # a rank died: the others would wait in the next collective for ever
                    q.terminate()
        time.sleep(0.05)
    if rc:
        sys.stderr.write("bench.py: a rank exited with status %d\n" % rc)
    return rc


# ------------------------------------------------------------------------------------------ workload
def make_cfgs(args):
    from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
    r, c, mn, mx = (int(x) for x in args.grid.split("x"))
    return (default_fe_cfg(grid_row=r, grid_col=c, grid_min=mn, grid_max=mx),
            default_ekf_cfg(max_cam_state_size=args.clones, compression_mode=COMPRESSION_MODES[args.compression]))


def sequence_seed(rank, world, u):
    """Seed of the u-th sequence of a rank.  Sequence 0 is the SENTINEL: the same seed on every rank, so that stream 0
    of every GPU computes the same thing and the cross-rank hash comparison means something; the others follow the
    sharding stream s -> rank s mod world: local sequence u is global stream u * world + rank."""
    return 0x5EED0000 if u == 0 else 0x5EED0000 + u * world + rank


def make_generators(oracle_py, args, rank, world):
    return [oracle_py.Synth(seed=sequence_seed(rank, world, u), width=args.width, height=args.height, n_static=25, n_loop=args.loop)
            for u in range(args.unique)]


def render_sequences(oracle_py, args, rank, world, n_keys):
    """Pre-render `unique` looping stereo sequences on the HOST (threads; the generator releases the GIL).  --host-render only:
    the default renders them on the device (msckf_stereo_c_amd/synth_device.py, identical bytes)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    syns = make_generators(oracle_py, args, rank, world)
    frames = np.empty((args.unique, 2, n_keys, args.height, args.width), np.uint8)

    def job(uk):
        u, k = uk
        a, b = syns[u].render(k)
        frames[u, 0, k] = a
        frames[u, 1, k] = b
        syns[u]._cache.clear()
    with ThreadPoolExecutor(max_workers=max(1, min(14, int(host_cores_available())))) as ex:
        list(ex.map(job, [(u, k) for u in range(args.unique) for k in range(n_keys)]))
    return syns, frames


def imu_array(syn, n):
    import numpy as np
    from msckf_stereo_c_amd.runner import IMU_SAMPLE
    out = np.zeros(n, IMU_SAMPLE)
    for j in range(n):
        s = syn.imu(j)
        out[j] = (s.time_stamp, tuple(s.angular_velocity), tuple(s.linear_acceleration))
    return out


def state_hash(ids, life, c0, c1, imu_state):
    """64-bit hash of what a stream has computed: live feature ids, lifetimes, cam0/cam1 pixels, the IMU state."""
    h = hashlib.blake2b(digest_size=8)
    for a in (ids, life, c0, c1, imu_state):
        h.update(memoryview(a).cast("B") if hasattr(a, "dtype") else bytes(a))
    return int.from_bytes(h.digest(), "little", signed=True)


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys) or None, os.cpu_count()


def quat_angle(qa, qb):
    """Rotation angle (rad) between unit quaternions given as rows [x y z w]; sign-insensitive."""
    import numpy as np
    d = np.abs(np.sum(qa * qb, axis=1))
    return 2.0 * np.arccos(np.clip(d, 0.0, 1.0))


def pose_compare(o_poses, g_poses):
    """GPU trajectory of a stream against the CPU oracle's over their common frames: max position / rotation difference
    and ATE RMSE (tools/ate_rmse.py: stamp association + Horn alignment, the reference README's evaluation)."""
    import numpy as np
    from tools.ate_rmse import ate_rmse
    m = min(len(o_poses), len(g_poses))
    if m < 3:
        return None
    o, g = o_poses[:m], g_poses[:m]
    if not np.array_equal(o["t"], g["t"]):
        return {"frames": int(m), "stamps_equal": False}
    dp = np.linalg.norm(o["p"] - g["p"], axis=1)
    dq = quat_angle(o["q"], g["q"])
    ate = ate_rmse(g["t"], g["p"], o["t"], o["p"])
    return {"frames": int(m), "stamps_equal": True, "max_dp_m": float(dp.max()), "max_dq_rad": float(dq.max()),
            "ate_rmse_m": float(ate["rmse"]), "path_length_m": float(np.linalg.norm(np.diff(o["p"], axis=0), axis=1).sum())}


def cpu_baseline(oracle_py, args, fe, ekf, total_gpu_frames, gpu_dump, gpu_poses=None):
    """The CPU oracle (kind "port": the reference itself cannot be built here) timed on this box's host
    cores, compiled like the reference (-O3, one thread per stream, headless).  Leg (i): one stream on one core (the
    reference's execution model).  Leg (ii): one stream per usable core, all at once.  Frames are rendered before the
    clock starts.  Leg (i) replays the sentinel sequence, so its feature ids at the frame the GPU stopped at are
    compared with stream 0 of the GPU run (id check inside the bench; the parity tests proper are tests/ -m gpu)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    usable = max(1, int(host_cores_available()))
    model, phys, logical = cpu_info()
    n_all = usable if args.cpu_all_frames > 0 else 1
    span = args.prime + max(args.cpu_frames, args.cpu_all_frames)
    check_at = total_gpu_frames if (gpu_dump is not None and args.prime < total_gpu_frames <= args.prime + args.cpu_frames) else None
    syns = [oracle_py.Synth(seed=sequence_seed(0, 1, u), width=args.width, height=args.height, n_static=25, n_loop=args.loop) for u in range(n_all)]
    with ThreadPoolExecutor(max_workers=usable) as ex:     # render outside the clock
        list(ex.map(lambda uk: syns[uk[0]].render(uk[1]), [(u, k) for u in range(n_all) for k in range(min(span, 25 + args.loop))]))
    # ---- (i) one stream, one core
    osys = oracle_py.OracleSystem(syns[0].calib, fe, ekf)
    syns[0].feed(osys, args.prime)
    id_check = None
    t0 = time.perf_counter()
    if check_at is not None:
        syns[0].feed(osys, check_at - args.prime, start=args.prime)
        o_ids, o_life, o_c0, o_c1, _ = osys.dump()
        g_ids, g_life, g_c0, g_c1 = gpu_dump
        id_check = {"frame": check_at, "features": int(len(o_ids)),
                    "ids_equal": bool(np.array_equal(o_ids, g_ids)), "lifetimes_equal": bool(np.array_equal(o_life, g_life)),
                    "pixels_equal": bool(np.array_equal(o_c0, g_c0) and np.array_equal(o_c1, g_c1))}
        if gpu_poses:
            # the metric's second half: trajectory of stream 0 of the timed batch against the oracle's, every frame up to here
            pc = pose_compare(osys.poses(), gpu_poses[0][:len(osys.poses())])
            if pc:
                id_check["pose_stream0"] = pc
        syns[0].feed(osys, args.prime + args.cpu_frames - check_at, start=check_at)
    else:
        syns[0].feed(osys, args.cpu_frames, start=args.prime)
    dt1 = time.perf_counter() - t0
    out = {"value": args.cpu_frames / dt1, "unit": "stereo frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of one %dx%d stream (grid %s, %d clones) after %d priming frames, CPU oracle (-O3, 1 thread), %.1f s"
                     % (args.cpu_frames, args.width, args.height, args.grid, args.clones, args.prime, dt1),
           "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "usable_cores": usable}
    # ---- (ii) one stream per usable core
    if args.cpu_all_frames > 0 and usable > 1:
        systems = [oracle_py.OracleSystem(s.calib, fe, ekf) for s in syns]
        with ThreadPoolExecutor(max_workers=usable) as ex:
            list(ex.map(lambda i: syns[i].feed(systems[i], args.prime), range(n_all)))
            t0 = time.perf_counter()
            list(ex.map(lambda i: syns[i].feed(systems[i], args.cpu_all_frames, start=args.prime), range(n_all)))
            dt = time.perf_counter() - t0
        out["all_cores"] = {"value": n_all * args.cpu_all_frames / dt, "unit": "stereo frames/s", "cores": n_all,
                            "sample": "%d streams x %d frames at once, one oracle thread per usable core, %.1f s" % (n_all, args.cpu_all_frames, dt)}
        if gpu_poses and id_check is not None:
            # these oracle streams replay sequences 0 .. n_all-1, which streams 0 .. n_all-1 of the GPU batch ran inside the
            # timed launches: compare the trajectories over the frames both have (filter state of the batch vs the oracle)
            worst = {"streams": 0, "frames": 0, "max_dp_m": 0.0, "max_dq_rad": 0.0, "ate_rmse_m": 0.0}
            for i in range(min(n_all, len(gpu_poses))):
                pc = pose_compare(systems[i].poses(), gpu_poses[i])
                if not pc or not pc.get("stamps_equal"):
                    continue
                worst["streams"] += 1
                worst["frames"] = max(worst["frames"], pc["frames"])
                for k in ("max_dp_m", "max_dq_rad", "ate_rmse_m"):
                    worst[k] = max(worst[k], pc[k])
            id_check["pose_batch"] = worst
    return out, id_check


def cgroup_cpu():
    """(cpu quota in cores or None, throttled microseconds so far or None) of this process's cgroup (v2)."""
    quota, thr = None, None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                thr = int(line.split()[1])
    except (OSError, ValueError):
        pass
    return quota, thr


def host_cores_available():
    """Cores this process may use: the cgroup quota if there is one, else the affinity mask."""
    quota, _ = cgroup_cpu()
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    return min(quota, aff) if quota else aff


# ------------------------------------------------------------------------------------------ one rank
def gather_hashes(h, world, device):
    import torch
    import torch.distributed as dist
    if world <= 1 or not dist.is_initialized():
        return [h]
    t = torch.tensor([h], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def rehearse(args, rank, world):
    """Multi-rank plumbing without device work (CPU tests): rendezvous, barrier, the three reductions, one JSON line."""
    import torch.distributed as dist
    from msckf_stereo_c_amd.dist_util import aggregate_throughput, shard_streams
    if world > 1:
        dist.init_process_group(backend="gloo")
    mine = shard_streams(args.streams * world, world, rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    elapsed, frames = aggregate_throughput(time.perf_counter() - t0, len(mine) * args.steps, world, device="cpu")
    hashes = gather_hashes(0x5EED if os.environ.get("MSKF_REHEARSE_CORRUPT_RANK") != str(rank) else 0xBAD, world, "cpu")
    if rank == 0:
        emit(json.dumps({"metric": "rehearsal of the rank plumbing (no device work, not a measurement)", "rehearsal": True,
                          "value": frames / elapsed, "unit": "stub frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed * 1e3 / args.steps, "streams_total": int(frames / args.steps),
                          "id_mismatch": sum(1 for h in hashes if h != hashes[0])}))
    if world > 1:
        dist.destroy_process_group()
    return 0


_T0 = time.perf_counter()


def progress(rank, what):
    """One stderr line per stage of the run (rank 0): where a run that stops making progress stopped."""
    if rank == 0:
        sys.stderr.write("bench.py [%7.2f s, monotonic %.3f] %s\n" % (time.perf_counter() - _T0, time.monotonic(), what))
        sys.stderr.flush()


_JSON_OUT = None


def keep_stdout_clean():
    """The contract is ONE JSON line on stdout: everything else that writes to file descriptor 1 (c10d's "[Gloo] Rank 0 is connected
    ..." line, RCCL / runtime chatter) is sent to stderr; the line itself goes out through a duplicate of the original descriptor."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _JSON_OUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)                    # nothing above has imported torch or touched HIP
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or let bench.py spawn them)\n" % (args.gpus, world))
        return 2
    keep_stdout_clean()
    if args.rehearse:
        return rehearse(args, rank, world)
    # Every group runs a front-end and a filter thread that wait for the GPU between phases.  Spinning in those waits is the
    # lower-latency choice while each thread has a core to itself; when the ranks of this node together run more such
    # threads than there are cores (a shared quota), parked waits leave the cores to the threads that have host work.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if "MSKF_WAIT" not in os.environ and 2 * args.groups * args.host_threads * local_world > host_cores_available():
        os.environ["MSKF_WAIT"] = "block"
    # The front-end of a group is one device call per frame, its thread mostly waits; the filter stage still has per-stream
    # host work between its device calls (observation tables, update descriptors): it gets a helper thread when the
    # cores are there (measured: 86.7 k -> 90.0 k stereo frames/s on the pool's 16-core share; round 3, with the device
    # the limit: two helpers shorten the host phases by a third and the device waits grow by as much, 99 k vs 103 k)
    ekf_host_threads = int(os.environ.get("MSKF_BENCH_EKF_HOST_THREADS", "0"))
    if not ekf_host_threads and args.host_threads == 1 and 2 * args.groups * local_world <= host_cores_available():
        ekf_host_threads = 3 if local_world == 1 and host_cores_available() >= 16 else 2
    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()      # rehearsal on fewer GPUs than ranks shares devices (gloo only)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from msckf_stereo_c_amd import build, runner as R
    build.build_all()
    from oracle import oracle_py            # synthetic generator + (rank 0) the cpu_baseline leg only
    oracle_py.build()

    fe, ekf = make_cfgs(args)
    per_group = max(1, args.streams // args.groups)
    n_groups = max(1, args.streams // per_group)
    n_streams = n_groups * per_group
    n_keys = 25 + args.loop
    if args.stagger < 0:
        args.stagger = (args.loop // n_groups + 1) if n_groups > 1 else 0
    if args.unique <= 0:
        args.unique = per_group
    shared_in_group = args.unique < per_group
    max_offset = args.stagger * (n_groups - 1) if args.unique < n_streams else 0
    total_frames = args.prime + args.warmup + args.steps

    syns = make_generators(oracle_py, args, rank, world)
    calib = syns[0].calib
    run = R.Runner(calib, fe, ekf, n_groups, per_group, device=local_rank, host_threads=args.host_threads, ekf_host_threads=ekf_host_threads)
    t_r0 = time.perf_counter()
    frame_bytes = args.width * args.height
    if args.host_render:
        _, frames = render_sequences(oracle_py, args, rank, world, n_keys)
        d_frames = None if args.host_images else torch.from_numpy(frames).cuda(local_rank)
        h_frames = torch.from_numpy(frames).pin_memory() if args.host_images else None
        del frames                                            # the host copy (GBs per rank) is not needed any more
    else:
        # the sequences are rendered ON THE DEVICE (synth_render.hip: the generator's own per-pixel source, one thread per four
        # pixels): start-up no longer scales with ranks x host cores (round 3: 52 s and 11.8 GB of host memory per rank).  The kernels
        # run on the first group's own stream: every stage of the pipeline has a hardware queue to itself (GPU_MAX_HW_QUEUES), a
        # generator on the null stream would bind a seventeenth
        from msckf_stereo_c_amd import synth_device
        d_frames = synth_device.render_sequences(syns, n_keys, torch.device("cuda", local_rank), hip_stream=run.hip_stream(0))
        h_frames = None
        if args.host_images:
            h_frames = torch.empty(d_frames.shape, dtype=torch.uint8).pin_memory()
            h_frames.copy_(d_frames)
            d_frames = None
    render_s = time.perf_counter() - t_r0
    progress(rank, "sequences rendered (%s), %d streams in %d batches" % ("host" if args.host_render else "device", n_streams, n_groups))
    if args.host_images:
        base, on_device = h_frames.data_ptr(), 0
    else:
        base, on_device = d_frames.data_ptr(), 2              # resident in HBM before the timed region, borrowed in place
    # pose_out.txt / path_ / points3d_ growth (msckf_vio.cpp:1296-1302, Q20) is off for the batch; the streams whose
    # trajectories are compared with the CPU oracle after the run keep theirs
    run.keep_trajectory(False)
    n_pose_streams = min(per_group, max(1, int(host_cores_available()))) if (rank == 0 and not args.no_cpu) else 1
    for s_ in range(n_pose_streams):
        run.keep_trajectory(True, s_)
    cooldown = (args.steps + 8) if n_groups > 1 else 0      # frames a group may step beyond its own W + K while the window is open
    if max_offset:
        run.set_stagger(args.stagger)
    hh_warm = 3
    hh_frames = (hh_warm + args.gram_steps + cooldown + 8) if args.gram_steps > 0 else 0      # frames the second window may add
    imus = [imu_array(s, (total_frames + max_offset + 2 * cooldown + hh_frames + 3) * 10 + 20) for s in syns]
    for s in range(n_streams):
        u = s % args.unique
        cam0 = base + (u * 2 + 0) * n_keys * frame_bytes
        cam1 = base + (u * 2 + 1) * n_keys * frame_bytes
        run.set_sequence(s, cam0, cam1, on_device, frame_bytes, 25, args.loop, 1403715273262142976, 50000000, imus[u])

    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    pipe = not args.no_pipeline
    run.run(0, args.prime, pipelined=pipe)                       # untimed: gravity/bias init, clone window fills, group offsets
    progress(rank, "primed: %d frames per stream" % args.prime)
    # per-kernel HIP events on every 5th launch of a kind (5 is coprime to the launches per step of every kind, so all of a
    # step's launches are sampled in turn); MSKF_BENCH_TIMING_PERIOD=1 times every launch, MSKF_BENCH_NO_KERNEL_TIMING=1 none
    timing_period = 0 if os.environ.get("MSKF_BENCH_NO_KERNEL_TIMING") else int(os.environ.get("MSKF_BENCH_TIMING_PERIOD", "5"))
    cpu_quota, thr0 = cgroup_cpu()
    if pipe:
        # W warm-up + K timed steps in ONE pipelined run: the group pipelines (front-end | filter threads) are not torn
        # down and refilled at the warm-up / timed boundary.  The groups' hardware queues are not served evenly (some
        # groups run a quarter of the run ahead of others), so the timed window is defined on the WORK: it opens when the
        # groups together have completed groups x W frames (front-end AND filter) and closes at groups x (W + K):
        # exactly K steps' worth of stream-frames are completed inside it, nothing is skipped, and every group is busy
        # from before it opens until after it closes (each runs at least W + K frames and keeps stepping until the
        # window is closed; those extra frames are real frames, finished but not counted).
        run.set_timing(timing_period)
        run.get_timing(reset=True)
        run.get_abi_host_time(reset=True)
        run.get_hostprof(reset=True)
        barrier()
        t_wall0 = time.perf_counter()
        elapsed = run.run_timed(args.prime, args.warmup, args.steps, max_extra=cooldown)
        barrier()
        wall_run = time.perf_counter() - t_wall0
        phases = run.get_window_phases()
    else:
        run.run(args.prime, args.warmup, pipelined=False)        # W untimed warmup steps
        run.set_timing(timing_period)
        run.get_timing(reset=True)
        run.get_phases(reset=True)
        run.get_abi_host_time(reset=True)
        barrier()
        t0 = time.perf_counter()
        run.get_hostprof(reset=True)
        run.run(args.prime + args.warmup, args.steps, pipelined=False)   # EXACTLY K timed steps
        barrier()
        elapsed = wall_run = time.perf_counter() - t0
        phases = run.get_phases(reset=True)
    progress(rank, "timed window done: %d steps in %.3f s" % (args.steps, elapsed))
    hostprof = run.get_hostprof()
    _, thr1 = cgroup_cpu()
    timing = run.get_timing(reset=True)
    abi_host = run.get_abi_host_time(reset=True)
    run.set_timing(False)
    windows = [run.window(g) for g in range(n_groups)] if pipe else []

    from msckf_stereo_c_amd.dist_util import aggregate_throughput
    elapsed, frames_total = aggregate_throughput(elapsed, n_streams * args.steps, world, device=red_dev)

    # what stream 0 (the sentinel sequence, group 0: no offset) had computed when its stages closed the timed window, i.e.
    # after frame prime + W + K - 1; identical on every rank by construction
    if pipe:
        s_ids, s_life, s_c0, s_c1, s_imu = run.mark_dump(0)
    else:
        s_ids, s_life, s_c0, s_c1, _ = run.dump(0)
        s_imu = run.imu_state(0)
    hashes = gather_hashes(state_hash(s_ids, s_life, s_c0, s_c1, s_imu), world, red_dev)
    id_mismatch = sum(1 for h in hashes if h != hashes[0])
    frames_by_group = [run.frames_done(g) - run.group_offset(g) - args.prime for g in range(n_groups)] if pipe else []

    # sanity of the workload actually processed (steady state reached, filter alive)
    feats = [len(run.dump(s)[0]) for s in range(0, n_streams, max(1, n_streams // 16))]
    n_feat = int(round(sum(feats) / len(feats)))
    n_clones = run.num_clones(0)
    n_upd = run.num_updates(0)

    status = 0
    n_tsqr, n_uncompressed, n_rows, n_resets = run.num_tsqr_updates(0), run.num_uncompressed_updates(0), run.stacked_rows(0), run.num_resets(0)

    # ---- second window: the same streams go on with the faster, not literal compression - Gram + regularised Cholesky (the MFMA
    #      GEMM pass), Householder only where the device asks for it - instead of the reference's own rule (Householder QR when the
    #      stack has more rows than columns, msckf_vio.cpp:795-821) that the headline window ran, so that the line carries both rates
    #      from one run of the driver's command (the trajectories checked against the oracle below run through both)
    hh = None
    if pipe and args.gram_steps > 0:
        run.set_compression(COMPRESSION_MODES["auto"])
        first2 = max(run.frames_done(g) - run.group_offset(g) for g in range(n_groups))      # every batch catches up to here first
        tsqr0 = run.num_tsqr_updates(0)
        run.set_timing(timing_period)
        run.get_timing(reset=True)
        barrier()
        el2 = run.run_timed(first2, hh_warm, args.gram_steps, max_extra=cooldown)
        barrier()
        timing2 = run.get_timing(reset=True)
        progress(rank, "second window (Gram + Cholesky) done: %d steps in %.3f s" % (args.gram_steps, el2))
        run.set_timing(False)
        run.set_compression(COMPRESSION_MODES[args.compression])
        el2, frames2 = aggregate_throughput(el2, n_streams * args.gram_steps, world, device=red_dev)
        dom2 = max(timing2, key=lambda k: timing2[k][0])
        hh = {"value_gram_cholesky": frames2 / el2, "compression": "auto", "steps": args.gram_steps, "warmup": hh_warm, "ms_per_step": el2 * 1e3 / args.gram_steps,
              "householder_updates_stream0_in_window_and_warmup": run.num_tsqr_updates(0) - tsqr0,
              "dominant_kernel": dom2, "kernels": {k: {"ms": round(v[0], 3), "launches": v[1], "avg_us": round(v[0] * 1e3 / max(v[1], 1), 2)}
                                                   for k, v in timing2.items() if v[1] and k.startswith("k_ekf")}}

    if rank == 0:
        value = frames_total / elapsed
        # ---- roofline of the dominant kernel (largest HIP-event time inside the timed region)
        dom = max(timing, key=lambda k: timing[k][0])
        # "k_ekf_feature_blocks" is the event span over up to five launches of an update (triangulation, pair blocks and the three
        # block classes): no single kernel of it runs longer than a third of the span.  It is the dominant KERNEL only when the
        # span is more than three times the longest single-kernel entry; otherwise that entry is.
        if dom == "k_ekf_feature_blocks":
            single = max((k for k in timing if k != dom), key=lambda k: timing[k][0])
            if timing[dom][0] < 3.0 * timing[single][0]:
                dom = single
        ms, launches, units = timing[dom]
        avg_s = max(ms * 1e-3 / max(launches, 1), 1e-12)
        if dom in ("k_ekf_feature_blocks", "k_ekf_gemm", "k_ekf_chol_lds", "k_ekf_trsm", "k_ekf_tsqr"):
            achieved = units / max(launches, 1) / avg_s / 1e12           # units = algorithmic FP64 flops
            roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None}
            if dom == "k_ekf_tsqr":
                roof["note"] = "a chain of Householder reflectors on the FP64 vector units, priced against the FP64 peak (vector = matrix = 78.6 TFLOP/s on MI355X)"
        else:
            # k_pyr_down (one launch for levels 1..3): level 0 read once + the three levels written = (64 / 21 + 1) bytes per output pixel
            per_unit = {"k_track4": LK_BYTES_PER_TRACK, "k_pyr_down": 64.0 / 21.0 + 1.0, "k_detect_cells": 1}.get(dom, 0)
            achieved = units / max(launches, 1) * per_unit / avg_s / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None}
        # HBM traffic per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
        # separately); only comparable when a launch covers the same number of streams
        try:
            pmc_all = json.load(open(PMC_TRAFFIC_FILE))
            pmc = pmc_all["kernels"].get(PROFILE_NAMES.get(dom, dom))
            spl = pmc_all.get("streams_per_launch", 96)
            if pmc and args.config == pmc_all.get("config", "c2"):
                # every stream of a launch reads its own images: traffic per launch is proportional to the streams in it
                roof["traffic"] = (pmc["fetch_kb_per_launch"] + pmc["write_kb_per_launch"]) * 1024.0 * per_group / spl
                roof["traffic_note"] = ("bytes/launch, raw FETCH_SIZE+WRITE_SIZE of %s (collected at %d streams per launch, scaled to %d; "
                                        "no gfx950 correction applied: dword loads, not the calibrated 16 B/lane stream)"
                                        % (os.path.relpath(PMC_TRAFFIC_FILE, ROOT), spl, per_group))
        except Exception:
            pass
        roof["avg_launch_us"] = avg_s * 1e6
        roof["units_per_launch"] = units / max(launches, 1)
        # HBM is not what bounds the track kernel (traffic is half its algorithmic bytes, it runs at a few per cent of the HBM
        # peak): its governing roof is VALU issue.  Instructions per wave and waves per launch from the committed SQ counter pass
        # (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES, collected at `streams_per_launch` streams, scaled to this run's launches).
        try:
            sq_all = json.load(open(PMC_SQ_FILE))
            sq = sq_all["kernels"].get(PROFILE_NAMES.get(dom, dom))
            spl = sq_all.get("streams_per_launch", 96)
            if sq and args.config == sq_all.get("config", "c2"):
                wave_insts = sq["valu_insts_per_wave"] * sq["waves_per_dispatch"] * per_group / spl
                roof["valu_issue"] = {"bound": "valu", "insts_per_wave": sq["valu_insts_per_wave"], "waves_per_launch": sq["waves_per_dispatch"] * per_group / spl,
                                      "achieved": wave_insts / avg_s, "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
                                      "frac": wave_insts / avg_s / VALU_ISSUE_PEAK, "source": os.path.relpath(PMC_SQ_FILE, ROOT)}
                if dom == "k_track4" and roof["units_per_launch"] > 0:
                    # wave-instructions per point track (a wavefront carries four points: round 2's 7026 per wave = 1757 per point)
                    roof["valu_issue"]["insts_per_point_track"] = wave_insts / roof["units_per_launch"]
        except Exception:
            pass
        kernels = {k: {"ms": round(v[0], 3), "launches": v[1], "avg_us": round(v[0] * 1e3 / max(v[1], 1), 2)} for k, v in timing.items() if v[1]}
        # FP64 matrix-core rate of the dense update kernels alone (the meaningful MFMA number is per kernel)
        gm = timing.get("k_ekf_gemm", (0, 0, 0))
        mfma = {"kernel": "k_ekf_gemm", "tflops": (gm[2] / (gm[0] * 1e-3) / 1e12) if gm[0] > 0 else None, "peak": FP64_PEAK_TFLOPS}
        if mfma["tflops"] is not None:
            mfma["frac"] = mfma["tflops"] / FP64_PEAK_TFLOPS
            mfma["note"] = "tflops / frac are over SURVEY 8(d)'s ALGORITHMIC flops (dense-d formulas); executed_* are what the matrix cores did"
        # executed flops: SQ_INSTS_VALU_MFMA_MOPS_F64 of the committed counter pass, per launch and mode, weighted by the modes'
        # dispatch counts, scaled from that pass's streams per launch to this run's; over the launch time measured in THIS run
        try:
            mm = json.load(open(PMC_MFMA_FILE))
            spl = mm.get("streams_per_launch", 96)
            modes = {k: v for k, v in mm["kernels"].items() if k.startswith("k_ekf_gemm")}
            nd = sum(v["dispatches"] for v in modes.values())
            if nd and gm[1] > 0 and args.config == mm.get("config", "c2"):
                gflop = sum(v["fp64_mfma_gflop_per_dispatch"] * v["dispatches"] for v in modes.values()) / nd * per_group / spl
                t_launch = gm[0] * 1e-3 / gm[1]
                mfma["executed_gflop_per_launch"] = gflop
                mfma["executed_tflops"] = gflop * 1e9 / t_launch / 1e12
                mfma["executed_frac"] = mfma["executed_tflops"] / FP64_PEAK_TFLOPS
                mfma["executed_alone"] = {k: {"tflops": v["tflops"], "mfma_busy_pct_simd": v["mfma_busy_pct_simd"]} for k, v in modes.items()}
                mfma["source"] = os.path.relpath(PMC_MFMA_FILE, ROOT)
        except Exception:
            pass
        seq_note = ("%d distinct looping sequences per GPU, one per stream of a group" % args.unique
                    + ("" if not max_offset else "; the %d groups replay them %d frames apart (never the same stereo pair at the same time)"
                       % (n_groups, args.stagger))
                    + ("; SHARED inside a launch" if shared_in_group else ""))
        # where a step's wall time goes on the two threads of a group (averages over groups, ms per step): each thread's items
        # add up to its own window (open -> close), which is ms_per_step up to the skew between the groups
        per = lambda names: {k: round(phases[k] * 1e3 / args.steps / n_groups, 3) for k in names}
        if pipe:
            # (a thread's window = the span between its first frame boundary inside the timed window and the first one after
            # it; summed over the groups and divided by K x groups it is that thread's wall time per group-step)
            fe_items, ekf_items = per(R.Runner.FE_THREAD_PHASES), per(R.Runner.EKF_THREAD_PHASES)
            host_phases = {
                "front_end_thread": dict(fe_items, sum=round(sum(fe_items.values()), 3),
                                         window=round(sum(w["fe_close"] - w["fe_open"] for w in windows) * 1e3 / args.steps / n_groups, 3)),
                "filter_thread": dict(ekf_items, sum=round(sum(ekf_items.values()), 3),
                                      window=round(sum(w["ekf_close"] - w["ekf_open"] for w in windows) * 1e3 / args.steps / n_groups, 3)),
                "frames_started_in_window": {"front_end": int(sum(w["fe_frames"] for w in windows)), "filter": int(sum(w["ekf_frames"] for w in windows)),
                                             "counted": args.steps * n_groups},
                # each >= W + K: a batch finishes its own frames and keeps stepping (drain) until the window has closed; how far
                # the batches were apart INSIDE the window is frames_completed_at_close_by_group (sums to groups x (W + K))
                "frames_run_by_group": frames_by_group,
                "frames_completed_at_close_by_group": [int(w["frames_at_close"]) for w in windows],
                "stage_ms_per_frame_by_group": {"front_end": [round((w["fe_close"] - w["fe_open"]) * 1e3 / max(w["fe_frames"], 1), 2) for w in windows],
                                                "filter": [round((w["ekf_close"] - w["ekf_open"]) * 1e3 / max(w["ekf_frames"], 1), 2) for w in windows]},
                "run_wall_ms": round(wall_run * 1e3, 1),
            }
            if os.environ.get("MSKF_BENCH_GROUP_PHASES"):
                host_phases["window_phases_ms_by_group"] = [{k: round(v * 1e3, 2) for k, v in run.get_window_phases_group(g).items() if v > 0} for g in range(n_groups)]
                host_phases["windows_ms_by_group"] = [{k: round((v - min(w["fe_open"] for w in windows if w["fe_open"] > 0)) * 1e3, 1) if k.endswith(("open", "close")) else v for k, v in w.items()} for w in windows]
        else:
            host_phases = per([k for k in phases if phases[k] > 0])
        out = {
            "metric": "stereo frames/sec/node on EuRoC-shape input; ATE RMSE vs CPU ref", "value": value, "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" (host-resident images, PCIe-inclusive)" if args.host_images else ""),
            "config": {"workload": "%s [%s]: %dx%d stereo, %d cam clones, grid %s (%d features/frame realised), 200 Hz IMU; "
                                   "KLT + EKF update on-GPU; %d independent streams per GPU batched per launch (%d host groups x %d threads%s); %s; "
                                   "pose_out.txt is not written and the reference's ever-growing path / point lists (Q20) are kept only for the %d streams whose trajectories "
                                   "are compared with the CPU oracle"
                                   % (CONFIGS[args.config]["name"], args.config, args.width, args.height, args.clones, args.grid, n_feat, n_streams,
                                      n_groups, args.host_threads, ", FE|EKF pipelined" if pipe else "", seq_note, n_pose_streams),
                       "streams_per_gpu": n_streams, "features_per_frame": n_feat, "cam_clones": n_clones,
                       "compression": args.compression, "ekf_updates_stream0": n_upd, "ekf_tsqr_updates_stream0": n_tsqr,
                       "ekf_uncompressed_updates_stream0": n_uncompressed,
                       "ekf_rows_per_update_stream0": round(n_rows / max(n_upd, 1), 1),
                       "ekf_resets_stream0": n_resets, "render_s": round(render_s, 1), "rendered_on": "host" if args.host_render else "device", "kernel_timing_period": timing_period,
                       "group_stagger_frames": args.stagger if max_offset else 0,
                       "host_cpu_quota": cpu_quota, "host_wait": os.environ.get("MSKF_WAIT", "spin"),
                       "filter_host_threads_per_group": ekf_host_threads or args.host_threads,
                       "scheduler": ("balanced: a group's queues and threads take, frame by frame, the batch of streams that is furthest behind"
                                     if (pipe and n_groups > 1) else "fixed: every batch of streams on its own group's queues"),
                       "host_throttled_ms_in_timed_region": None if thr0 is None or thr1 is None else round((thr1 - thr0) / 1e3, 1)},
            "id_mismatch": id_mismatch,
            "value_gram_cholesky": hh["value_gram_cholesky"] if hh else None, "gram_window": hh,
            "roofline": roof, "mfma": mfma, "kernels": kernels,
            "host_bookkeeping_us_per_stream_frame": {k: round(v * 1e6 / (n_streams * args.steps), 2) for k, v in hostprof.items()},
            "host_phases_ms_per_step": host_phases,
            "abi_host_ms_per_step": {k: round(v * 1e3 / args.steps / n_groups, 3) for k, v in abi_host.items()},
        }
        if not args.no_cpu and world == 1 and args.cpu_frames > 0:
            for s in syns:
                s._cache.clear()
            del syns
            gpu_poses = [run.poses(s_) for s_ in range(n_pose_streams)]
            out["cpu_baseline"], out["id_check_vs_oracle"] = cpu_baseline(oracle_py, args, fe, ekf, total_frames, (s_ids, s_life, s_c0, s_c1), gpu_poses)
            chk = out["id_check_vs_oracle"] or {}
            pcs = [chk[k] for k in ("pose_stream0", "pose_batch") if k in chk and chk[k].get("stamps_equal", True)]
            if pcs:
                # BASELINE.json north_star: poses within 1e-4 m / 1e-4 rad of the CPU reference path
                out["ate_rmse_vs_cpu_ref_m"] = max(p_["ate_rmse_m"] for p_ in pcs)
                out["pose_err_vs_cpu_ref"] = {"max_dp_m": max(p_["max_dp_m"] for p_ in pcs), "max_dq_rad": max(p_["max_dq_rad"] for p_ in pcs),
                                              "streams": max(p_.get("streams", 1) for p_ in pcs), "tolerance": 1e-4}
                out["pose_within_tolerance"] = bool(out["pose_err_vs_cpu_ref"]["max_dp_m"] <= 1e-4 and out["pose_err_vs_cpu_ref"]["max_dq_rad"] <= 1e-4)
        # a run whose results are wrong has no headline: every check that was made must hold, or the exit status says so
        failed = []
        if id_mismatch:
            failed.append("id_mismatch=%d (sentinel stream differs across ranks)" % id_mismatch)
        chk = out.get("id_check_vs_oracle") or {}
        for k in ("ids_equal", "lifetimes_equal", "pixels_equal"):
            if k in chk and not chk[k]:
                failed.append("id_check_vs_oracle.%s" % k)
        for k in ("pose_stream0", "pose_batch"):
            if k in chk and chk[k].get("stamps_equal") is False:
                failed.append("id_check_vs_oracle.%s.stamps_equal" % k)
        if out.get("pose_within_tolerance") is False:
            failed.append("pose_within_tolerance %s" % json.dumps(out["pose_err_vs_cpu_ref"]))
        out["checks_failed"] = failed
        emit(json.dumps(out))
        if failed:
            sys.stderr.write("bench.py: RESULT CHECKS FAILED: %s\n" % "; ".join(failed))
            status = 1
    run.close()
    if world > 1:
        # every rank leaves with rank 0's verdict (a launcher sees one status)
        t = torch.tensor([status], dtype=torch.int32, device=red_dev)
        dist.broadcast(t, src=0)
        status = int(t.item())
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
