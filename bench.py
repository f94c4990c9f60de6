#!/usr/bin/env python3
"""bench.py — throughput of the MI355X hot path (stereo KLT front-end + MSCKF update) on synthetic
EuRoC-shaped streams.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 launched by torch.distributed.run, one rank per GPU, RCCL only for the
barrier and the max/sum aggregation: the streams are independent units, no data-path collective).
A "step" is one stereo frame of EVERY stream of the batch: full front-end (pyramids, detector, temporal
LK, stereo LK, gates) + back-end (IMU propagation, augmentation, triangulation, Jacobians, gating, QR,
Kalman update, clone pruning).  Inputs (rendered stereo pairs) are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per HIP stream (8 groups x {front-end, filter}); the ROCm default of 4 multiplexes the
# 16 streams and serialises unrelated kernels.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

# SURVEY.md §8(d) algorithmic work per unit
LK_BYTES_PER_TRACK = 4 * (17 * 17 + 16 * 16)      # 4 levels x (17x17 template + 16x16 search footprint) u8
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6                           # FP64 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MSKF_BENCH_STREAMS", "768")), help="VIO streams per GPU")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("MSKF_BENCH_GROUPS", "8")), help="host thread groups per GPU")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("MSKF_BENCH_HOST_THREADS", "1")), help="host threads per group")
    ap.add_argument("--no-pipeline", action="store_true", help="run front-end and filter of a group in lockstep on one thread")
    ap.add_argument("--unique", type=int, default=0, help="distinct rendered sequences per GPU (streams cycle over them); "
                    "0 = one per stream of a group, so that no two streams of a launch read the same image (no L2 sharing)")
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--clones", type=int, default=30)
    ap.add_argument("--grid", type=str, default="8x10x4x5", help="rows x cols x min x max features per cell")
    ap.add_argument("--prime", type=int, default=75, help="untimed frames before warmup: gravity init + clone window fill")
    ap.add_argument("--loop", type=int, default=100, help="frames per trajectory period")
    ap.add_argument("--cpu-frames", type=int, default=250, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("MSKF_BENCH_BACKEND", "nccl"), help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--host-images", action="store_true", help="stereo pairs stay in host memory: PCIe-inclusive rate (not the headline value)")
    return ap.parse_args()


def make_cfgs(args):
    from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg
    r, c, mn, mx = (int(x) for x in args.grid.split("x"))
    return default_fe_cfg(grid_row=r, grid_col=c, grid_min=mn, grid_max=mx), default_ekf_cfg(max_cam_state_size=args.clones)


def render_sequences(oracle_py, args, rank, n_keys):
    """Pre-render `unique` looping stereo sequences on the host (threads; the generator releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    syns = [oracle_py.Synth(seed=0x5EED0000 + 64 * rank + u, width=args.width, height=args.height, n_static=25, n_loop=args.loop)
            for u in range(args.unique)]
    frames = np.empty((args.unique, 2, n_keys, args.height, args.width), np.uint8)

    def job(uk):
        u, k = uk
        a, b = syns[u].render(k)
        frames[u, 0, k] = a
        frames[u, 1, k] = b
        syns[u]._cache.clear()
    with ThreadPoolExecutor(max_workers=min(14, os.cpu_count() or 8)) as ex:
        list(ex.map(job, [(u, k) for u in range(args.unique) for k in range(n_keys)]))
    return syns, frames


def imu_array(syn, n):
    from msckf_stereo_c_amd.runner import IMU_SAMPLE
    out = np.zeros(n, IMU_SAMPLE)
    for j in range(n):
        s = syn.imu(j)
        out[j] = (s.time_stamp, tuple(s.angular_velocity), tuple(s.linear_acceleration))
    return out


def cpu_baseline(oracle_py, syn, fe, ekf, prime, frames):
    """The CPU oracle (a port: the reference itself cannot be built here) on ONE stream, one core."""
    osys = oracle_py.OracleSystem(syn.calib, fe, ekf)
    syn.feed(osys, prime)
    t0 = time.perf_counter()
    syn.feed(osys, frames, start=prime)
    dt = time.perf_counter() - t0
    return frames / dt, dt


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


def main():
    args = parse()
    # Every group runs a front-end and a filter thread that wait for the GPU between phases.  Spinning in those waits is the
    # lower-latency choice while each thread has a core to itself; when the ranks of this node together run more such
    # threads than there are cores (a shared quota), parked waits leave the cores to the threads that have host work.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if "MSKF_WAIT" not in os.environ and 2 * args.groups * args.host_threads * local_world > host_cores_available():
        os.environ["MSKF_WAIT"] = "block"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    total_frames = args.prime + args.warmup + args.steps
    if args.unique <= 0:
        args.unique = min(per_group, 128)

    t_r0 = time.perf_counter()
    syns, frames = render_sequences(oracle_py, args, rank, n_keys)
    render_s = time.perf_counter() - t_r0
    frame_bytes = args.width * args.height
    if args.host_images:
        h_frames = torch.from_numpy(frames).pin_memory()
        base, on_device = h_frames.data_ptr(), 0
        del frames
    else:
        d_frames = torch.from_numpy(frames).cuda(local_rank)  # resident in HBM before the timed region
        base, on_device = d_frames.data_ptr(), 2              # borrowed in place
        del frames                                            # the host copy (GBs per rank) is not needed any more
    calib = syns[0].calib
    run = R.Runner(calib, fe, ekf, n_groups, per_group, device=local_rank, host_threads=args.host_threads)
    run.keep_trajectory(False)
    imus = [imu_array(s, (total_frames + 3) * 10 + 20) for s in syns]
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
    run.run(0, args.prime, pipelined=pipe)                       # untimed: gravity/bias init, clone window fills, steady state
    run.run(args.prime, args.warmup, pipelined=pipe)             # W untimed warmup steps
    run.set_timing(not os.environ.get("MSKF_BENCH_NO_KERNEL_TIMING"))
    run.get_timing(reset=True)
    run.get_phases(reset=True)
    barrier()
    cpu_quota, thr0 = cgroup_cpu()
    t0 = time.perf_counter()
    run.get_hostprof(reset=True)
    run.run(args.prime + args.warmup, args.steps, pipelined=pipe)   # EXACTLY K timed steps
    barrier()
    elapsed = time.perf_counter() - t0
    hostprof = run.get_hostprof()
    _, thr1 = cgroup_cpu()
    timing = run.get_timing(reset=True)
    phases = run.get_phases(reset=True)
    run.set_timing(False)

    from msckf_stereo_c_amd.dist_util import aggregate_throughput
    elapsed, frames_total = aggregate_throughput(elapsed, n_streams * args.steps, world, device=red_dev)

    # sanity of the workload actually processed (steady state reached, filter alive)
    n_feat = len(run.dump(0)[0])
    n_clones = run.num_clones(0)
    n_upd = run.num_updates(0)

    if rank == 0:
        value = frames_total / elapsed
        # ---- roofline of the dominant kernel (largest HIP-event time inside the timed region)
        dom = max(timing, key=lambda k: timing[k][0])
        ms, launches, units = timing[dom]
        avg_s = max(ms * 1e-3 / max(launches, 1), 1e-12)
        if dom in ("k_ekf_feature_blocks", "k_ekf_gemm", "k_ekf_chol_lds", "k_ekf_trsm"):
            achieved = units / max(launches, 1) / avg_s / 1e12           # units = algorithmic FP64 flops
            roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None}
        else:
            per_unit = {"k_lk_points": LK_BYTES_PER_TRACK, "k_pyr_down": 5, "k_detect_cells": 1}.get(dom, 0)
            achieved = units / max(launches, 1) * per_unit / avg_s / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None}
        # HBM traffic per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
        # separately, profiles/r01_pmc_hbm_traffic.json); only comparable when a launch covers the same number of streams
        try:
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
            pmc = pmc_all["kernels"].get(dom)
            if pmc and per_group == pmc_all.get("streams_per_launch", 64):
                roof["traffic"] = (pmc["fetch_kb_per_launch"] + pmc["write_kb_per_launch"]) * 1024.0
                roof["traffic_note"] = "bytes/launch, raw FETCH_SIZE+WRITE_SIZE of profiles/r01_pmc_hbm_traffic.json (no gfx950 correction applied)"
        except Exception:
            pass
        roof["avg_launch_us"] = avg_s * 1e6
        roof["units_per_launch"] = units / max(launches, 1)
        kernels = {k: {"ms": round(v[0], 3), "launches": v[1], "avg_us": round(v[0] * 1e3 / max(v[1], 1), 2)} for k, v in timing.items()}
        out = {
            "metric": "stereo frames/sec/node on EuRoC-shape input", "value": value, "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" (host-resident images, PCIe-inclusive)" if args.host_images else ""),
            "config": {"workload": "Single MI355X: %dx%d stereo, %d cam clones, grid %s (%d features/frame), 200 Hz IMU; "
                                   "KLT + EKF update on-GPU; %d independent streams per GPU batched per launch (%d host groups x %d threads%s)"
                                   % (args.width, args.height, args.clones, args.grid, n_feat, n_streams, n_groups, args.host_threads,
                                      ", FE|EKF pipelined" if pipe else ""),
                       "streams_per_gpu": n_streams, "features_per_frame": n_feat, "cam_clones": n_clones,
                       "ekf_updates_stream0": n_upd, "ekf_resets_stream0": run.num_resets(0), "render_s": round(render_s, 1),
                       "host_cpu_quota": cpu_quota, "host_wait": os.environ.get("MSKF_WAIT", "spin"),
                       "host_throttled_ms_in_timed_region": None if thr0 is None or thr1 is None else round((thr1 - thr0) / 1e3, 1)},
            "roofline": roof, "kernels": kernels,
            "host_bookkeeping_us_per_stream_frame": {k: round(v * 1e6 / (n_streams * args.steps), 2) for k, v in hostprof.items()},
            "host_phases_ms_per_step": {k: round(v * 1e3 / args.steps / n_groups, 3) for k, v in phases.items()},
        }
        if not args.no_cpu and world == 1 and args.cpu_frames > 0:
            fps, dt = cpu_baseline(oracle_py, syns[0], fe, ekf, args.prime, args.cpu_frames)
            out["cpu_baseline"] = {"value": fps, "unit": "stereo frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of one %dx%d stream after %d priming frames, CPU oracle (-O3, 1 thread), %.1f s"
                                             % (args.cpu_frames, args.width, args.height, args.prime, dt)}
        print(json.dumps(out))
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
