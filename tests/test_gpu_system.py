"""End-to-end GPU parity: the product path (host mirror classes -> C-ABI -> HIP kernels) against the CPU
oracle on identical seeded synthetic EuRoC-shaped streams, in the reference harness call order.

Bar (BASELINE.json north_star): feature ids and pixels bit-exact per frame; poses within 1e-4 m / 1e-4 rad.
"""
import numpy as np
import pytest

from msckf_stereo_c_amd import runner as R
from msckf_stereo_c_amd.ctypes_types import default_ekf_cfg, default_fe_cfg

pytestmark = pytest.mark.gpu

PYR_LAUNCHES_PER_FRAME = 1      # k_pyr_down3: levels 1..3 of both cameras of every stream in one launch
POS_TOL = 1e-4   # metres
ANG_TOL = 1e-4   # radians


def quat_angle(q1, q2):
    d = abs(float(np.dot(q1, q2)))
    return 2.0 * np.arccos(min(1.0, d))


def compare_frame(k, osys, run, stream=0):
    o_ids, o_life, o_c0, o_c1, o_info = osys.dump()
    g_ids, g_life, g_c0, g_c1, g_info = run.dump(stream)
    assert np.array_equal(o_ids, g_ids), "frame %d: ids differ" % k
    assert np.array_equal(o_life, g_life), "frame %d: lifetimes differ" % k
    assert np.array_equal(o_c0, g_c0) and np.array_equal(o_c1, g_c1), "frame %d: pixels differ" % k
    assert (o_info.before_tracking, o_info.after_tracking, o_info.after_matching, o_info.after_ransac) == \
           (g_info.before_tracking, g_info.after_tracking, g_info.after_matching, g_info.after_ransac)


def compare_msgs(osys, run, stream=0):
    om, gm = osys.msg(), run.msg(stream)
    assert len(om) == len(gm)
    for f in ("id", "u0", "v0", "u1", "v1"):
        assert np.array_equal(om[f], gm[f]), f


def compare_poses(osys, run, stream=0):
    op, gp = osys.poses(), run.poses(stream)
    assert len(op) == len(gp) and len(op) > 0
    assert np.array_equal(op["t"], gp["t"])
    dp = np.linalg.norm(op["p"] - gp["p"], axis=1)
    da = np.array([quat_angle(a, b) for a, b in zip(op["q"], gp["q"])])
    assert dp.max() < POS_TOL, "max position difference %g m" % dp.max()
    assert da.max() < ANG_TOL, "max orientation difference %g rad" % da.max()
    return dp.max(), da.max()


@pytest.mark.parametrize("w,h,n_frames,seed", [(376, 240, 110, 0x5EED0000), (752, 480, 60, 0x5EED0001)])
def test_single_stream_matches_oracle(oracle, w, h, n_frames, seed):
    syn = oracle.Synth(seed=seed, width=w, height=h)
    fe, ekf = default_fe_cfg(), default_ekf_cfg()
    osys = oracle.OracleSystem(syn.calib, fe, ekf)
    run = R.Runner(syn.calib, fe, ekf, 1, 1)
    view = R.StreamView(run)
    checked = []

    def on_frame(k, _):
        compare_frame(k, osys, run)
        checked.append(k)

    syn.feed(osys, n_frames)   # oracle first (keeps its own per-frame dump only for the last frame) ...
    # ... so replay both in lockstep instead: fresh oracle, frame by frame
    osys = oracle.OracleSystem(syn.calib, fe, ekf)
    j = 0
    for k in range(n_frames):
        t_img = syn.frame_time(k)
        while True:
            s = syn.imu(j)
            j += 1
            osys.imu(s)
            view.imu(s)
            if not (s.time_stamp <= t_img):
                break
        a, b = syn.render(k)
        osys.stereo(a, b, t_img)
        view.stereo(a, b, t_img)
        osys.backend()
        view.backend()
        compare_frame(k, osys, run)
        if k % 10 == 0:
            compare_msgs(osys, run)
    compare_msgs(osys, run)
    assert osys.num_updates() == run.num_updates() and (osys.num_updates() > 0 or n_frames < 45)
    assert osys.num_resets() == run.num_resets() == 0
    dp, da = compare_poses(osys, run)
    Po, Pg = osys.cov(), run.cov()
    assert Po.shape == Pg.shape
    perr = np.abs(Po - Pg).max() / np.abs(Po).max()
    print("frames %d: max |dp| %.3e m, max dtheta %.3e rad, max |dP|/max|P| %.3e" % (n_frames, dp, da, perr))
    assert perr < 1e-5
    so, sg = osys.imu_state(), run.imu_state()
    assert np.allclose(so, sg, atol=1e-5)
    run.close()


def test_batched_streams_match_their_oracles(oracle):
    """Three different streams stepped in one batch (one launch per phase) == three independent oracle runs."""
    w, h, n_frames = 376, 240, 70
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=12)
    syns = [oracle.Synth(seed=0x5EED0010 + i, width=w, height=h, motion_scale=0.8 + 0.2 * i) for i in range(3)]
    osys = [oracle.OracleSystem(s.calib, fe, ekf) for s in syns]
    run = R.Runner(syns[0].calib, fe, ekf, 1, 3)
    cur = [0, 0, 0]
    for k in range(n_frames):
        imgs0, imgs1, ts = [], [], []
        for i, syn in enumerate(syns):
            t_img = syn.frame_time(k)
            while True:
                s = syn.imu(cur[i])
                cur[i] += 1
                osys[i].imu(s)
                run.imu(i, s)
                if not (s.time_stamp <= t_img):
                    break
            a, b = syn.render(k)
            osys[i].stereo(a, b, t_img)
            osys[i].backend()
            imgs0.append(a); imgs1.append(b); ts.append(t_img)
        run.step(imgs0, imgs1, ts)
        for i in range(3):
            compare_frame(k, osys[i], run, i)
    for i in range(3):
        compare_msgs(osys[i], run, i)
        compare_poses(osys[i], run, i)
        assert osys[i].num_updates() == run.num_updates(i) > 0
    run.close()


def _attach_sequences(oracle, run, syns, n_frames, keep):
    from msckf_stereo_c_amd.runner import IMU_SAMPLE
    for i, syn in enumerate(syns):
        n_keys = syn.n_static + syn.n_loop
        frames = np.empty((2, n_keys, syn.h, syn.w), np.uint8)
        for k in range(min(n_keys, n_frames + 1)):
            a, b = syn.render(k)
            frames[0, k], frames[1, k] = a, b
        imu = np.zeros((n_frames + 3) * 10 + 20, IMU_SAMPLE)
        for j in range(len(imu)):
            s = syn.imu(j)
            imu[j] = (s.time_stamp, tuple(s.angular_velocity), tuple(s.linear_acceleration))
        keep.append(frames)
        fb = syn.w * syn.h
        run.set_sequence(i, frames.ctypes.data, frames.ctypes.data + n_keys * fb, 0, fb, syn.n_static, syn.n_loop,
                         1403715273262142976, 50000000, imu)


def test_pipelined_run_is_identical_to_lockstep(oracle):
    """BatchGroup::run_pipelined (front-end thread | filter thread, two HIP streams) == lockstep run == oracle."""
    w, h, n_frames = 376, 240, 70
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syns = [oracle.Synth(seed=0x5EED0020 + i, width=w, height=h) for i in range(2)]
    keep = []
    runs = []
    for pipelined in (False, True):
        run = R.Runner(syns[0].calib, fe, ekf, 1, 2, host_threads=1)
        _attach_sequences(oracle, run, syns, n_frames, keep)
        run.run(0, n_frames, threaded=True, pipelined=pipelined)
        runs.append(run)
    a, b = runs
    for i in range(2):
        for x, y in zip(a.dump(i)[:4], b.dump(i)[:4]):
            assert np.array_equal(x, y)
        pa, pb = a.poses(i), b.poses(i)
        assert len(pa) == len(pb) > 20
        assert np.array_equal(pa["p"], pb["p"]) and np.array_equal(pa["q"], pb["q"])     # bitwise: same arithmetic, same order
        assert np.array_equal(a.cov(i), b.cov(i))
        assert a.num_updates(i) == b.num_updates(i) > 0
        # and both equal the oracle fed in the reference harness order
        osys = oracle.OracleSystem(syns[i].calib, fe, ekf)
        syns[i].feed(osys, n_frames)
        assert np.array_equal(osys.dump()[0], b.dump(i)[0])
        op = osys.poses()
        assert np.abs(op["p"] - pb["p"]).max() < POS_TOL
    for r in runs:
        r.close()


def test_half_batches_on_a_shared_stream_are_identical(oracle):
    """halves=2: a group drives two staggered half-batches per stage, each on its own context sharing the stage's
    HIP stream (mskf_ctx_create_shared), through the *_batch_begin / *_batch_end halves of the C-ABI.  Everything must be
    bit-identical to the one-batch run, the filter included: which kernels handle a stream's update is decided per
    STREAM (EkfStreamDev::route), never from the rest of the batch, so a stream's arithmetic does not depend on which
    streams share its launches.  (Round 2 chose the route per batch and this test had to be relaxed to 1e-9.)"""
    w, h, n_frames = 376, 240, 60
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syns = [oracle.Synth(seed=0x5EED0060 + i, width=w, height=h) for i in range(4)]
    keep, runs = [], []
    for halves in (1, 2):
        run = R.Runner(syns[0].calib, fe, ekf, 1, 4, host_threads=1, halves=halves)
        _attach_sequences(oracle, run, syns, n_frames, keep)
        run.run(0, n_frames, threaded=True, pipelined=True)
        runs.append(run)
    a, b = runs
    for i in range(4):
        for x, y in zip(a.dump(i)[:4], b.dump(i)[:4]):
            assert np.array_equal(x, y)
        pa, pb = a.poses(i), b.poses(i)
        assert len(pa) == len(pb) > 15
        assert np.array_equal(pa["p"], pb["p"]) and np.array_equal(pa["q"], pb["q"])
        assert np.array_equal(a.cov(i), b.cov(i))
        assert a.num_updates(i) == b.num_updates(i) > 0
    for r in runs:
        r.close()


def _lockstep(oracle, syn, fe, ekf, n_frames, frame_hook=None, check_every=1, after_frame=None):
    """Run oracle and GPU path in lockstep on (possibly modified) frames of `syn`; returns (osys, run)."""
    osys = oracle.OracleSystem(syn.calib, fe, ekf)
    run = R.Runner(syn.calib, fe, ekf, 1, 1)
    view = R.StreamView(run)
    j = 0
    for k in range(n_frames):
        t_img = syn.frame_time(k)
        while True:
            s = syn.imu(j)
            j += 1
            osys.imu(s)
            view.imu(s)
            if not (s.time_stamp <= t_img):
                break
        a, b = syn.render(k)
        if frame_hook:
            a, b = frame_hook(k, a, b)
        osys.stereo(a, b, t_img)
        view.stereo(a, b, t_img)
        osys.backend()
        view.backend()
        if k % check_every == 0:
            compare_frame(k, osys, run)
        if after_frame:
            after_frame(k, osys, run)
    return osys, run


def test_compat_switches_off(oracle):
    """compat_flags = 0: cleared message (no Q1), true previous timestamp (no Q2), sieve-order responses (no Q4) and the
    2-point RANSAC of image_processor.cpp:911-1135 on both cameras (no Q5).  Since round 4 the RANSAC runs
    inside the device frame (fe_book1): every frame after the first is one device call."""
    syn = oracle.Synth(seed=0x5EED0030, width=376, height=240)
    fe, ekf = default_fe_cfg(compat=0), default_ekf_cfg()
    osys, run = _lockstep(oracle, syn, fe, ekf, 60)
    assert run.num_device_frames() == 59
    compare_msgs(osys, run)
    assert len(run.msg()) == len(run.dump()[0])          # message holds exactly the live features
    compare_poses(osys, run)
    run.close()


def test_ransac_rejects_independently_moving_patch(oracle):
    """RANSAC on (compat_flags = 0): a textured patch drifting across both images carries features whose temporal flow
    contradicts the camera motion; they pass the stereo gate and must be dropped by the 2-point RANSAC — in the same
    frames, with the same survivors (ids, pixels, counters), by the GPU path and the oracle."""
    syn = oracle.Synth(seed=0x5EED0033, width=376, height=240)
    fe, ekf = default_fe_cfg(compat=0), default_ekf_cfg()
    rng = np.random.default_rng(5)
    patch = (rng.integers(0, 2, (12, 12)) * 160 + 40).astype(np.uint8).repeat(5, axis=0).repeat(5, axis=1)   # 60 x 60 blocks

    def hook(k, a, b):
        if k < 30:
            return a, b
        a, b = a.copy(), b.copy()
        x = 40 + 4 * (k - 30)
        y = 60 + 2 * (k - 30)
        if x + 60 + 8 < a.shape[1] and y + 60 < a.shape[0]:
            a[y:y + 60, x + 8:x + 68] = patch                # 8 px of disparity: a consistent stereo pair
            b[y:y + 60, x:x + 60] = patch
        return a, b

    rejected = []

    def after(k, osys, run):
        info = run.dump()[4]
        assert info.after_ransac <= info.after_matching
        rejected.append(info.after_matching - info.after_ransac)

    osys, run = _lockstep(oracle, syn, fe, ekf, 70, frame_hook=hook, after_frame=after)
    assert run.num_device_frames() == 69                     # the RANSAC ran on the device, between the track kernels
    assert sum(rejected[:30]) <= 2                           # static scene: (almost) nothing to reject
    assert sum(rejected[31:]) >= 5, rejected                 # features riding on the patch are thrown out
    compare_msgs(osys, run)
    compare_poses(osys, run)
    run.close()


def test_textureless_stream(oracle):
    """Empty inputs end to end: flat images give no detections, no tracks, no updates — and no crash."""
    syn = oracle.Synth(seed=0x5EED0031, width=376, height=240)
    flat = np.full((240, 376), 90, np.uint8)
    osys, run = _lockstep(oracle, syn, default_fe_cfg(), default_ekf_cfg(), 40, frame_hook=lambda k, a, b: (flat, flat))
    assert len(run.dump()[0]) == 0 and len(run.msg()) == 0
    assert run.num_updates() == osys.num_updates() == 0
    compare_poses(osys, run)                               # pure IMU dead reckoning, identical host arithmetic
    run.close()


def test_blackout_and_recovery(oracle):
    """All features lost at once (3 flat frames mid-sequence): one large update (row cap), then re-detection."""
    syn = oracle.Synth(seed=0x5EED0032, width=376, height=240)
    flat = np.full((240, 376), 128, np.uint8)
    fe, ekf = default_fe_cfg(grid_row=6, grid_col=8, grid_min=3, grid_max=4), default_ekf_cfg(max_cam_state_size=12)

    def hook(k, a, b):
        return (flat, flat) if 45 <= k < 48 else (a, b)
    osys, run = _lockstep(oracle, syn, fe, ekf, 75, frame_hook=hook)
    compare_msgs(osys, run)
    ids = run.dump()[0]
    assert len(ids) > 80 and ids.min() > 100                # everything alive was born after the blackout
    assert osys.num_updates() == run.num_updates() > 10
    compare_poses(osys, run)
    run.close()


def test_trajectory_error_against_ground_truth(oracle):
    """SURVEY §8(d) metric 2: ATE RMSE (Horn alignment, tools/ate_rmse.py) of the GPU trajectory against the synthetic
    ground truth is the oracle's (centimetres), and against the oracle's trajectory it is ~0."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("ate_rmse", os.path.join(os.path.dirname(__file__), "..", "tools", "ate_rmse.py"))
    ate = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ate)
    syn = oracle.Synth(width=376, height=240)
    n = 110
    osys, run = _lockstep(oracle, syn, default_fe_cfg(), default_ekf_cfg(), n, check_every=10)
    gp, op = run.poses(0), osys.poses()
    assert len(gp) == len(op) > 80
    k0 = n - len(gp)
    gt_t = np.array([syn.frame_time(k) for k in range(k0, n)])
    gt_p = np.array([syn.gt_pose(k)["p"] for k in range(k0, n)])
    r_gt = ate.ate_rmse(gp["t"], gp["p"], gt_t, gt_p, max_dt=1e-3)
    r_or = ate.ate_rmse(op["t"], op["p"], gt_t, gt_p, max_dt=1e-3)
    r_x = ate.ate_rmse(gp["t"], gp["p"], op["t"], op["p"], max_dt=1e-6)
    assert r_gt["pairs"] == len(gp)
    assert r_gt["rmse"] < 0.03 and abs(r_gt["rmse"] - r_or["rmse"]) < 1e-6
    assert r_x["rmse"] < 1e-6
    run.close()


def test_stress_shape_c3(oracle):
    """SURVEY §8 config C3 image shape: 1280x720, 10x20 grid (~1000 features per frame); a 24-clone window so that the
    pruning update with several hundred 2-observation features runs inside the 45 frames (the d = 321 / 381 algebra of
    C3 / C5 is covered by test_gpu_kernels.py::test_ekf_update_many_clones).  Lockstep with the oracle."""
    syn = oracle.Synth(seed=0x5EED00C3, width=1280, height=720)
    fe = default_fe_cfg(grid_row=10, grid_col=20, grid_min=4, grid_max=5)
    ekf = default_ekf_cfg(max_cam_state_size=24)
    osys, run = _lockstep(oracle, syn, fe, ekf, 48, check_every=6)
    assert len(run.dump()[0]) > 600
    compare_msgs(osys, run)
    assert osys.num_updates() == run.num_updates() > 5
    compare_poses(osys, run)
    run.close()


def test_bench_scale_replicas_are_bit_identical(oracle):
    """Size-independent property at the benchmark's shape (752x480, 30-clone window, 8x10 grid, several pipelined groups
    of batched streams): streams fed the same sequence are bit-identical whatever their slot, group and thread — ids,
    pixels, poses and covariance — through steady state with lost-feature and pruning updates; one of them is checked
    against the oracle."""
    w, h, n_frames = 752, 480, 80
    fe = default_fe_cfg(grid_row=8, grid_col=10, grid_min=4, grid_max=5)
    ekf = default_ekf_cfg(max_cam_state_size=30)
    uniq = [oracle.Synth(seed=0x5EED0040 + i, width=w, height=h) for i in range(2)]
    n_groups, per_group = 3, 6
    run = R.Runner(uniq[0].calib, fe, ekf, n_groups, per_group, host_threads=1)
    keep = []
    syns = [uniq[s % 2] for s in range(n_groups * per_group)]
    _attach_sequences(oracle, run, syns, n_frames, keep)
    run.run(0, n_frames, threaded=True, pipelined=True)
    names = ("ids", "lifetimes", "cam0 pixels", "cam1 pixels")
    for s in range(2, n_groups * per_group):
        ref = s % 2
        for nm, x, y in zip(names, run.dump(ref)[:4], run.dump(s)[:4]):
            assert np.array_equal(x, y), "stream %d differs from its replica %d in %s" % (s, ref, nm)
        pa, pb = run.poses(ref), run.poses(s)
        assert len(pa) == len(pb) > 40, "stream %d: %d poses, replica %d: %d" % (s, len(pb), ref, len(pa))
        if not (np.array_equal(pa["p"], pb["p"]) and np.array_equal(pa["q"], pb["q"])):
            bad = np.nonzero(np.any(pa["p"] != pb["p"], axis=1) | np.any(pa["q"] != pb["q"], axis=1))[0]
            raise AssertionError("stream %d pose differs from replica %d first at pose %d of %d, max |dp| %.3e"
                                 % (s, ref, bad[0], len(pa), np.abs(pa["p"] - pb["p"]).max()))
        dP = np.abs(run.cov(ref) - run.cov(s)).max()
        assert dP == 0.0, "stream %d covariance differs from replica %d by %.3e" % (s, ref, dP)
        assert run.num_updates(ref) == run.num_updates(s) > 20
    assert run.num_clones(0) >= 28 and len(run.dump(0)[0]) > 250
    osys = oracle.OracleSystem(uniq[0].calib, fe, ekf)
    uniq[0].feed(osys, n_frames)
    assert np.array_equal(osys.dump()[0], run.dump(0)[0])
    op, gp = osys.poses(), run.poses(0)
    assert np.abs(op["p"] - gp["p"]).max() < POS_TOL
    run.close()


def test_first_updates_identical_across_busy_groups(oracle):
    """Regression: the stacked-Jacobian buffer is allocated at a stream's first update; its zero fill used to go to the
    null stream, which is not ordered against the (non-blocking) context streams, and on a busy device it could land on
    top of the feature kernel's rows — the first update of a few streams then silently did nothing.  Many groups running
    freely through the first updates (frames 26-33) must leave every replica bit-identical, on several fresh runners."""
    w, h, n_frames = 752, 480, 34
    fe = default_fe_cfg(grid_row=8, grid_col=10, grid_min=4, grid_max=5)
    ekf = default_ekf_cfg(max_cam_state_size=30)
    uniq = [oracle.Synth(seed=0x5EED0050 + i, width=w, height=h) for i in range(2)]
    n_groups, per_group = 6, 12
    from msckf_stereo_c_amd.runner import IMU_SAMPLE
    packs = []          # the replicas share the rendered frames of their sequence
    for syn in uniq:
        n_keys = syn.n_static + syn.n_loop
        frames = np.empty((2, n_keys, h, w), np.uint8)
        for k in range(n_frames + 1):
            frames[0, k], frames[1, k] = syn.render(k)
        imu = np.zeros((n_frames + 3) * 10 + 20, IMU_SAMPLE)
        for j in range(len(imu)):
            m = syn.imu(j)
            imu[j] = (m.time_stamp, tuple(m.angular_velocity), tuple(m.linear_acceleration))
        packs.append((frames, imu, n_keys, syn))
    for attempt in range(4):
        run = R.Runner(uniq[0].calib, fe, ekf, n_groups, per_group, host_threads=1)
        for s in range(n_groups * per_group):
            frames, imu, n_keys, syn = packs[s % 2]
            run.set_sequence(s, frames.ctypes.data, frames.ctypes.data + n_keys * w * h, 0, w * h, syn.n_static, syn.n_loop,
                             1403715273262142976, 50000000, imu)
        run.run(0, n_frames, threaded=True, pipelined=False)
        assert run.num_updates(0) >= 3
        for s in range(2, n_groups * per_group):
            a, b = run.cov(s % 2), run.cov(s)
            assert a.shape == b.shape and np.array_equal(a, b), "attempt %d: stream %d covariance differs from its replica" % (attempt, s)
            assert np.array_equal(run.poses(s % 2)["p"], run.poses(s)["p"])
        run.close()


def test_c3_full_size(oracle):
    """BASELINE configs[2] (C3 of SURVEY §8) at full size, everything together: 1280x720, the 10x20x(4..5) grid
    (~900-1000 features per frame) AND a 50-clone window (d = 321) that fills and prunes, so the gate matrices (up to
    200 rows) and the 300-column factorisations run on the global-memory paths inside a real sequence.  Ids and pixels
    bit-exact, poses against the oracle at the usual bar."""
    syn = oracle.Synth(seed=0x5EED0C33, width=1280, height=720)
    fe = default_fe_cfg(grid_row=10, grid_col=20, grid_min=4, grid_max=5)
    ekf = default_ekf_cfg(max_cam_state_size=50)
    osys, run = _lockstep(oracle, syn, fe, ekf, 86, check_every=5)
    assert run.num_clones() >= 48 and len(run.dump()[0]) > 800
    assert osys.num_updates() == run.num_updates() > 20
    compare_msgs(osys, run)
    compare_poses(osys, run)
    run.close()


def test_c5_4k_60_clones(oracle):
    """BASELINE configs[4] (C5 of SURVEY §8) shape: 3840x2160 stereo, 20x25x(4..5) grid (~1700-2000 features per frame),
    max_cam_state_size = 60 (d = 381).  92 frames in lockstep with the oracle: the window fills to 60 clones and prunes,
    lost-feature updates hit the 1500-row cap.  Feature ids, lifetimes and pixels bit-exact in every frame, poses at
    1e-4 m / 1e-4 rad.  The frames are rendered ahead by a thread pool (the generator is the slow part at 4K)."""
    from concurrent.futures import ThreadPoolExecutor
    n_frames = 92
    syn = oracle.Synth(seed=0x5EED00C5, width=3840, height=2160)
    fe = default_fe_cfg(grid_row=20, grid_col=25, grid_min=4, grid_max=5)
    ekf = default_ekf_cfg(max_cam_state_size=60)
    import os
    workers = max(2, min(14, len(os.sched_getaffinity(0))))
    twins = [oracle.Synth(seed=0x5EED00C5, width=3840, height=2160) for _ in range(workers)]     # one generator per thread
    rendered = {}

    def job(k):
        t = twins[k % workers]
        a, b = t.render(k)
        t._cache.clear()
        return k, a, b
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for k, a, b in ex.map(job, range(n_frames)):
            rendered[k] = (a, b)
    syn.render = lambda k: rendered[k]
    osys, run = _lockstep(oracle, syn, fe, ekf, n_frames, check_every=1)
    assert run.num_clones() >= 58 and len(run.dump()[0]) > 1500
    assert osys.num_updates() == run.num_updates() > 30
    compare_msgs(osys, run)
    compare_poses(osys, run)
    Po, Pg = osys.cov(), run.cov()
    assert Po.shape == Pg.shape == (21 + 6 * run.num_clones(),) * 2
    assert np.abs(Po - Pg).max() / np.abs(Po).max() < 1e-5
    run.close()


def test_long_run_c2_grid(oracle):
    """260 frames of one stream at the benchmark's own shape (752x480, 8x10x(5..6) grid ~ 440 features, 30 clones) in
    lockstep with the oracle: ~240 filter frames with a lost-feature update (compressed, uncompressed when it stacks no
    more rows than active columns) and a pruning update (thread-per-feature blocks, fused small update) every other
    frame.  Ids, lifetimes and pixels bit-exact in every frame, poses at 1e-4 m / 1e-4 rad, covariance at 1e-5."""
    syn = oracle.Synth(seed=0x5EED0C22, width=752, height=480, n_static=25, n_loop=100)
    fe = default_fe_cfg(grid_row=8, grid_col=10, grid_min=5, grid_max=6)
    ekf = default_ekf_cfg(max_cam_state_size=30)
    osys, run = _lockstep(oracle, syn, fe, ekf, 260, check_every=1)
    assert len(run.dump()[0]) >= 400 and run.num_clones() >= 28
    assert osys.num_updates() == run.num_updates() > 300
    assert run.num_uncompressed_updates() > 5               # both shapes of the lost-feature update occurred
    compare_msgs(osys, run)
    dp, da = compare_poses(osys, run)
    Po, Pg = osys.cov(), run.cov()
    perr = np.abs(Po - Pg).max() / np.abs(Po).max()
    print("260 frames: max |dp| %.3e m, max dtheta %.3e rad, |dP|/|P| %.3e, %d updates (%d uncompressed, %d TSQR)"
          % (dp, da, perr, run.num_updates(), run.num_uncompressed_updates(), run.num_tsqr_updates()))
    assert perr < 1e-5
    run.close()


def test_staggered_groups_run_ahead(oracle):
    """MultiRunner::set_stagger (bench.py: replicas of a looping sequence in different groups are kept frames apart so
    that they never read the same stereo pair at the same time): group g is g * delta frames ahead, and each group's
    stream equals the oracle fed that many frames."""
    w, h, n_frames, delta = 376, 240, 40, 7
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syn = oracle.Synth(seed=0x5EED0060, width=w, height=h)
    keep = []
    run = R.Runner(syn.calib, fe, ekf, 3, 1, host_threads=1)
    _attach_sequences(oracle, run, [syn, syn, syn], n_frames + 2 * delta, keep)
    run.set_stagger(delta)
    run.run(0, 25, threaded=True, pipelined=True)
    run.run(25, n_frames - 25, threaded=True, pipelined=True)
    for g in range(3):
        assert run.group_offset(g) == g * delta
        osys = oracle.OracleSystem(syn.calib, fe, ekf)
        syn.feed(osys, n_frames + g * delta)
        for x, y in zip(osys.dump()[:4], run.dump(g)[:4]):
            assert np.array_equal(x, y)
        op, gp = osys.poses(), run.poses(g)
        assert len(op) == len(gp) and np.abs(op["p"] - gp["p"]).max() < POS_TOL
    run.close()


def test_balanced_runner_with_fewer_workers_than_batches(oracle):
    """MultiRunner::run_balanced: batches of streams are not tied to queues - a front-end worker takes the batch that is
    furthest behind, a filter worker the oldest handed-off frame.  With ONE front-end worker and TWO filter workers for three
    batches (Runner.set_workers) every batch is run by borrowed contexts most of the time; each must still
    equal the oracle fed the same frames (ids and pixels bit-exact, poses within tolerance), and the streams must be back
    on their own contexts afterwards (a second, ordinary run continues them)."""
    w, h, n_frames, delta = 376, 240, 44, 4
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syn = oracle.Synth(seed=0x5EED0063, width=w, height=h)
    keep = []
    run = R.Runner(syn.calib, fe, ekf, 3, 1, host_threads=1)
    _attach_sequences(oracle, run, [syn, syn, syn], n_frames + 2 * delta + 8, keep)
    run.set_stagger(delta)
    run.set_workers(1, 2)
    run.run(0, 30, threaded=True, pipelined=True)
    run.set_workers(0, 0)
    run.run(30, n_frames - 30, threaded=True, pipelined=True)
    for g in range(3):
        osys = oracle.OracleSystem(syn.calib, fe, ekf)
        syn.feed(osys, n_frames + g * delta)
        for x, y in zip(osys.dump()[:4], run.dump(g)[:4]):
            assert np.array_equal(x, y)
        op, gp = osys.poses(), run.poses(g)
        assert len(op) == len(gp) and np.abs(op["p"] - gp["p"]).max() < POS_TOL
    run.close()


def test_nap_wait_mode_in_a_child_process():
    """MSKF_WAIT=nap (the completion mark is polled between short sleeps instead of spun on) and MSKF_WAIT=block (parked on a
    blocking-sync event): the mode is read once per process, so each runs tests/child_wait_mode.py - three batches through
    the balanced runner against the oracle - in a child of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("nap:25", "block"):
        env = dict(os.environ, MSKF_WAIT=mode)
        res = subprocess.run([sys.executable, os.path.join(root, "tests", "child_wait_mode.py")], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        assert ("OK wait=%s" % mode) in res.stdout


_TIMED_RUN = {}


def _timed_run(oracle):
    """ONE run of MultiRunner::run_timed (three staggered batches), shared by the two tests below: what was computed, and
    how the window was accounted."""
    if _TIMED_RUN:
        return _TIMED_RUN
    w, h, prime, warm, steps, delta, extra = 376, 240, 26, 3, 9, 5, 12
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syn = oracle.Synth(seed=0x5EED0061, width=w, height=h)
    keep = []
    run = R.Runner(syn.calib, fe, ekf, 3, 1, host_threads=1)
    _attach_sequences(oracle, run, [syn, syn, syn], prime + warm + steps + 2 * delta + extra + 2, keep)
    run.set_stagger(delta)
    run.run(0, prime, threaded=True, pipelined=True)
    run.set_timing(1)
    run.get_timing(reset=True)
    elapsed = run.run_timed(prime, warm, steps, max_extra=extra)
    timing = run.get_timing(reset=True)
    run.set_timing(False)
    _TIMED_RUN.update(dict(shape=(prime, warm, steps, delta, extra), fe=fe, ekf=ekf, syn=syn, elapsed=elapsed, timing=timing,
                           phases=run.get_window_phases(), windows=[run.window(g) for g in range(3)],
                           done=[run.frames_done(g) - run.group_offset(g) for g in range(3)],
                           marks=[run.mark_dump(g) for g in range(3)], dumps=[run.dump(g) for g in range(3)],
                           poses=[run.poses(g) for g in range(3)]))
    run.close()
    return _TIMED_RUN


def test_timed_window_results(oracle):
    """MultiRunner::run_timed (bench.py): warm-up + timed steps in ONE pipelined run.  The window is defined on the work
    (it opens when the batches together have completed n_groups x warm-up frames and closes at n_groups x (warm-up + steps)),
    every batch runs at least warm-up + steps frames and keeps stepping until the window is closed.  The sentinel snapshot
    taken after frame warm-up + steps of a batch equals the oracle at exactly that frame whatever the batch did afterwards,
    the live state has moved on by the drain frames, and the trajectory over all frames stays within tolerance."""
    T = _timed_run(oracle)
    prime, warm, steps, delta, extra = T["shape"]
    syn, fe, ekf = T["syn"], T["fe"], T["ekf"]
    assert T["elapsed"] > 0
    for g in range(3):
        done = T["done"][g]
        assert prime + warm + steps <= done <= prime + warm + steps + extra
        # the batch's sentinel after frame prime + warm + steps (+ offset) = the oracle after exactly that many frames
        osys = oracle.OracleSystem(syn.calib, fe, ekf)
        syn.feed(osys, prime + warm + steps + g * delta)
        ids, life, c0, c1, imu = T["marks"][g]
        o = osys.dump()
        assert np.array_equal(o[0], ids) and np.array_equal(o[1], life) and np.array_equal(o[2], c0) and np.array_equal(o[3], c1)
        assert np.abs(osys.imu_state() - imu).max() < POS_TOL
        # ... and the live state has moved on by the frames the batch stepped while the window was still open
        syn.feed(osys, done - (prime + warm + steps), start=prime + warm + steps + g * delta)
        for x, y in zip(osys.dump()[:4], T["dumps"][g][:4]):
            assert np.array_equal(x, y)
        op, gp = osys.poses(), T["poses"][g]
        assert len(op) == len(gp) and np.abs(op["p"] - gp["p"]).max() < POS_TOL


def test_timed_window_accounting(oracle):
    """The accounting of the same run, asserted structurally (nothing here depends on how fast the box is): the window
    holds exactly steps x batches completed frames; every stage opened its gates before it closed them; every phase item
    is non-negative and a thread's items do not add up to more than its own window; kernels are timed only inside the
    window (the pyramid launches equal the front-end frames started in it)."""
    T = _timed_run(oracle)
    prime, warm, steps, delta, extra = T["shape"]
    wins, ph = T["windows"], T["phases"]
    # completed frames of the run when the window closed: warm-up + timed steps of every batch, wherever each batch stood
    assert sum(int(wd["frames_at_close"]) for wd in wins) == 3 * (warm + steps)
    for g, wd in enumerate(wins):
        assert 0 < wd["fe_open"] < wd["fe_close"] and 0 < wd["ekf_open"] < wd["ekf_close"]
        assert 0 <= wd["frames_at_close"] <= T["done"][g] - prime
        assert wd["fe_frames"] >= 1 and wd["ekf_frames"] >= 1
    fe_frames = sum(int(wd["fe_frames"]) for wd in wins)
    assert fe_frames <= 3 * (steps + extra + 2)
    assert all(v >= 0.0 for v in ph.values())
    fe_sum = sum(ph[k] for k in R.Runner.FE_THREAD_PHASES)
    fe_win = sum(wd["fe_close"] - wd["fe_open"] for wd in wins)
    ekf_sum = sum(ph[k] for k in R.Runner.EKF_THREAD_PHASES)
    ekf_win = sum(wd["ekf_close"] - wd["ekf_open"] for wd in wins)
    assert 0 < fe_sum <= fe_win * (1 + 1e-6) + 1e-6 and 0 < ekf_sum <= ekf_win * (1 + 1e-6) + 1e-6
    # kernels are timed only inside the window: PYR_LAUNCHES_PER_FRAME pyramid launches per front-end frame started in it
    assert T["timing"]["k_pyr_down"][1] == PYR_LAUNCHES_PER_FRAME * fe_frames


def test_stream_results_do_not_depend_on_the_batch(oracle):
    """One stream (a) alone and (b) as a member of a batch of DIFFERENT streams — other scenes, another start phase, so
    the batch mixes lost-feature updates of every size class with pruning updates and empty updates — over 300 frames:
    identical feature ids / pixels, bit-identical poses in every frame and a bit-identical covariance, i.e. every gate
    decision of every update fell the same way (a flipped gate changes the stacked rows and with them P)."""
    w, h, n_frames = 376, 240, 300
    fe, ekf = default_fe_cfg(), default_ekf_cfg(max_cam_state_size=10)
    syn = oracle.Synth(seed=0x5EED0070, width=w, height=h)
    others = [oracle.Synth(seed=0x5EED0071 + i, width=w, height=h, motion_scale=0.5 + 0.5 * i) for i in range(5)]
    keep = []
    alone = R.Runner(syn.calib, fe, ekf, 1, 1, host_threads=1)
    _attach_sequences(oracle, alone, [syn], n_frames, keep)
    batch = R.Runner(syn.calib, fe, ekf, 1, 6, host_threads=1)
    _attach_sequences(oracle, batch, others[:2] + [syn] + others[2:], n_frames, keep)
    alone.run(0, n_frames, threaded=True, pipelined=True)
    batch.run(0, n_frames, threaded=True, pipelined=True)
    for x, y in zip(alone.dump(0)[:4], batch.dump(2)[:4]):
        assert np.array_equal(x, y)
    pa, pb = alone.poses(0), batch.poses(2)
    assert len(pa) == len(pb) > 250
    assert np.array_equal(pa["p"], pb["p"]) and np.array_equal(pa["q"], pb["q"])
    assert np.array_equal(alone.cov(0), batch.cov(2))
    assert alone.num_updates(0) == batch.num_updates(2) > 100
    assert np.array_equal(alone.imu_state(0), batch.imu_state(2))
    alone.close()
    batch.close()


def test_device_books_equal_host_books(oracle):
    """The front-end's bookkeeping on the device (mskf_fe_frame_batch_*: fe_book1 / fe_book2 between the track kernels,
    the grid resident in device memory) against the phased path with the books on the host (mskf_fe_track + the host
    mirror's phaseAfter1 / phaseAfter2), on the same streams: identical grids in every frame, identical messages
    (Q1 tail included), identical tracking info, poses within rounding — and both equal the oracle.  Streams: a normal
    one, a fast one (many lost features), a grid with partial rows / columns (333 x 251, quirk Q7), and one with the
    reference's quirks off, i.e. with the 2-point RANSAC between the tracks (host: cg::two_point_ransac after the first track
    call; device: inside fe_book1)."""
    cases = [(376, 240, 0x5EED0080, 1.0, default_fe_cfg()), (376, 240, 0x5EED0081, 2.5, default_fe_cfg(grid_row=3, grid_col=4, grid_min=2, grid_max=3)),
             (333, 251, 0x5EED0082, 1.0, default_fe_cfg(grid_row=4, grid_col=5, grid_min=3, grid_max=4)),
             (376, 240, 0x5EED0083, 1.5, default_fe_cfg(compat=0))]       # RANSAC on: host mirror's twoPointRansac vs fb_two_point_ransac in the kernel
    for w, h, seed, motion, fe in cases:
        ekf = default_ekf_cfg(max_cam_state_size=10)
        syn = oracle.Synth(seed=seed, width=w, height=h, motion_scale=motion)
        osys = oracle.OracleSystem(syn.calib, fe, ekf)
        runs = []
        for host in (1, 0):
            R.set_fe_books_on_host(host)
            runs.append(R.Runner(syn.calib, fe, ekf, 1, 1))
        try:
            j = 0
            for k in range(60):
                t_img = syn.frame_time(k)
                while True:
                    s = syn.imu(j)
                    j += 1
                    osys.imu(s)
                    for r in runs:
                        r.imu(0, s)
                    if not (s.time_stamp <= t_img):
                        break
                a, b = syn.render(k)
                osys.stereo(a, b, t_img)
                osys.backend()
                for host, r in zip((1, 0), runs):
                    R.set_fe_books_on_host(host)
                    r.step([a], [b], [t_img])
                compare_frame(k, osys, runs[0])
                compare_frame(k, osys, runs[1])
                compare_msgs(osys, runs[1])
            compare_poses(osys, runs[1])
            pa, pb = runs[0].poses(0), runs[1].poses(0)
            assert np.array_equal(pa["p"], pb["p"]) and np.array_equal(pa["q"], pb["q"])
        finally:
            R.set_fe_books_on_host(-1)
            for r in runs:
                r.close()
