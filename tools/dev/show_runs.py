"""Print the headline, per-kernel averages and stage times of bench JSON lines (gpurun_out/<name>.json ...)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    r = d["roofline"]
    print("%s: %.1f f/s, %.2f ms/step, failed=%s, dom %s frac %.4f (%.0f us), render %.1fs, hh=%s" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d.get("checks_failed"), r["kernel"], r["frac"], r.get("avg_launch_us", 0),
        d["config"].get("render_s", -1), round(d.get("value_gram_cholesky") or d.get("value_householder") or 0) or None))
    print("   kernels", {k.replace("k_ekf_", "e_").replace("k_", ""): (round(v["avg_us"]), v["launches"]) for k, v in d["kernels"].items()})
    hp = d["host_phases_ms_per_step"]
    if "filter_thread" in hp:
        print("   frames", hp["frames_run_by_group"], hp.get("frames_completed_at_close_by_group"))
        print("   fe ", {k: v for k, v in hp["front_end_thread"].items() if v > 0.05})
        print("   ekf", {k: v for k, v in hp["filter_thread"].items() if v > 0.05})
    w2 = d.get("gram_window") or d.get("householder_window")
    if w2:
        print("   hh ", {k.replace("k_ekf_", ""): v["avg_us"] for k, v in w2["kernels"].items()}, w2["dominant_kernel"])
