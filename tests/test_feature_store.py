"""Host bookkeeping tables of the filter (csrc/host/feature_store.h) against the reference's std::map semantics:
a randomised C++ driver performs the filter's operations (observe, lose, prune two clones, stale ids coming back,
duplicate records) on both and compares ids, order, masks and every observation value after every frame."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_feature_store_matches_std_map(tmp_path):
    exe = str(tmp_path / "feature_store_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "feature_store_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ")
