"""Build recipe: compiles the hand-written gfx950 kernels + C-ABI into in-tree shared libraries.

    python -m msckf_stereo_c_amd.build          # libmskf_hip.so (+ libmskf_host.so)

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the arithmetic
contract: the device and host point math must not be FMA-contracted.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# extra compiler flags for experiments (e.g. MSKF_EXTRA_FLAGS=-DEKF_NO_PRIO python -m msckf_stereo_c_amd.build --force)
COMMON += [f for f in os.environ.get("MSKF_EXTRA_FLAGS", "").split() if f]

HIP_SRCS = [
    "hip/fe_kernels.hip",
    "hip/ekf_kernels.hip",
    "hip/ekf_linalg.hip",
    "abi/mskf_capi_fe.cpp",
    "abi/mskf_capi_ekf.cpp",
]
HOST_SRCS = [
    "host/image_processor.cpp",
    "host/msckf_vio.cpp",
    "host/system.cpp",
    "host/batch_runner.cpp",
    "host/host_capi.cpp",
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _all_sources():
    deps = []
    for root, _, files in os.walk(CSRC):
        deps += [os.path.join(root, f) for f in files]
    inc = os.path.join(HERE, "..", "include")
    deps += [os.path.join(inc, f) for f in os.listdir(inc)]
    return deps


def build_hip(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    target = os.path.join(OUT, "libmskf_hip.so")
    if not force and not _newer(target, _all_sources()):
        return target
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared"] + COMMON + ["-o", target] + [os.path.join(CSRC, s) for s in HIP_SRCS]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


def build_host(force=False, verbose=False):
    """Host mirror of the reference classes (cg::ImageProcessor / MsckfVio / System) above the C-ABI."""
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HOST_SRCS if os.path.exists(os.path.join(CSRC, s))]
    if not srcs:
        return None
    target = os.path.join(OUT, "libmskf_host.so")
    if not force and not _newer(target, _all_sources()):
        return target
    build_hip(force=False, verbose=verbose)
    cmd = ["g++", "-shared", "-pthread"] + COMMON + ["-o", target] + srcs + ["-L" + OUT, "-lmskf_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


def build_synth_hip(force=False, verbose=False):
    """Device renderer of the synthetic bench / test sequences (csrc/synth/synth_render.hip): an input generator, a library of
    its own, not part of the C-ABI."""
    os.makedirs(OUT, exist_ok=True)
    target = os.path.join(OUT, "libmskf_synth_hip.so")
    srcs = [os.path.join(CSRC, "synth", "synth_render.hip"), os.path.join(CSRC, "synth", "synth.h")]
    if not force and not _newer(target, srcs):
        return target
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared"] + COMMON + ["-o", target, srcs[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


def build_app(force=False, verbose=False):
    """Headless counterpart of the reference harness apps/run_euroc_single_thread.cpp (zlib PNG reader)."""
    src = os.path.join(CSRC, "apps", "run_euroc_single_thread.cpp")
    target = os.path.join(OUT, "run_euroc_single_thread")
    if not force and not _newer(target, _all_sources()):
        return target
    cmd = ["g++"] + COMMON + ["-o", target, src, "-L" + OUT, "-lmskf_host", "-lmskf_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


class _BuildLock:
    """Exclusive advisory lock on the build directory: the ranks of a multi-process launch all call build_all()
    at start-up; one of them (re)builds, the others wait and then find the libraries up to date."""

    def __enter__(self):
        import fcntl
        os.makedirs(OUT, exist_ok=True)
        self.f = open(os.path.join(OUT, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def build_all(force=False, verbose=False):
    with _BuildLock():
        a = build_hip(force, verbose)
        b = build_host(force, verbose)
        build_app(force, verbose)
        build_synth_hip(force, verbose)
    return a, b


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
