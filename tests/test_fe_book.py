"""The front-end's device bookkeeping (msckf_stereo_c_amd/csrc/hip/fe_book.h) executed on the CPU: the header is written so
that the same source runs on the host (phases of independent items, no atomics, no cross-lane operations); the C++ harness
runs fe_book1 / fe_book2 over random multi-frame scenarios and compares every frame bit for bit with a restatement of the
reference's own flow (image_processor.cpp:416-513, :622-768) on std::map / std::stable_sort.  Every other trial runs with the
2-point RANSAC of :482-500 between the tracks (the reference side calls the product's host implementation of twoPointRansac,
which tests/test_ransac.py pins to the oracle; the device side runs fb_two_point_ransac inside fe_book1)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fe_book_matches_reference_flow(tmp_path):
    exe = str(tmp_path / "fe_book_test")
    from msckf_stereo_c_amd import build
    _, host = build.build_all()
    libdir = os.path.dirname(host)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-variable", "-I", ROOT, "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "fe_book_test.cpp"), "-L" + libdir, "-lmskf_host", "-lmskf_hip", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe, "400"], text=True)
    assert "device logic == reference flow" in out, out
