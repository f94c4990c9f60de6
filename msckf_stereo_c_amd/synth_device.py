"""Device renderer of the synthetic stereo sequences (csrc/synth/synth_render.hip) — bench / test INPUT GENERATOR, not part of
the VIO path.  The per-image parameters (camera pose of the frame) and the per-camera ray tables come from the host generator
(the Synth wrapper of the test infrastructure: render_params / ray_table); every pixel is then evaluated on the GPU by the same source the host renderer runs
(synth::shade_pixel), so the bytes are identical to Synth.render (tests/test_gpu_kernels.py::test_device_renderer_equals_host)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "_build", "libmskf_synth_hip.so")
        if not os.path.exists(p):
            raise RuntimeError("libmskf_synth_hip.so is not built (python -m msckf_stereo_c_amd.build)")
        L = C.CDLL(p)
        L.synth_hip_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.synth_hip_render.restype = C.c_int
        L.synth_hip_render_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.synth_hip_render_host.restype = C.c_int
        _LIB = L
    return _LIB


def render_sequences(syns, n_keys, device, hip_stream=None):
    """Frames 0 .. n_keys - 1 of both cameras of every generator in `syns` (same image size), rendered on `device`: a uint8 torch
    tensor [len(syns), 2, n_keys, h, w] resident in HBM.  hip_stream: the HIP stream (an integer handle) the kernels run on; a
    bench that gives every stage of its pipeline a hardware queue of its own passes one of those, so the generator binds no
    further queue (None = the null stream)."""
    import torch
    w, h = syns[0].w, syns[0].h
    assert (w * h) % 4 == 0, "the device renderer stores four pixels per thread"
    out = torch.empty((len(syns), 2, n_keys, h, w), dtype=torch.uint8, device=device)
    # (calibration only: the same for every seed; through pinned memory - see synth_hip_render about pageable sources)
    rays = [torch.from_numpy(syns[0].ray_table(c)).pin_memory().to(device) for c in (0, 1)]
    torch.cuda.synchronize(device)
    from .ctypes_types import RENDER_IMG
    with torch.cuda.device(device):
        for u, s in enumerate(syns):
            assert (s.w, s.h) == (w, h)
            params = np.zeros(2 * n_keys, RENDER_IMG)
            for c in (0, 1):
                for k in range(n_keys):
                    params[c * n_keys + k] = s.render_params(k, c)
            rc = lib().synth_hip_render(params.ctypes.data, len(params), rays[0].data_ptr(), rays[1].data_ptr(), w, h, out[u].data_ptr(), w * h, hip_stream)
            if rc != 0:
                raise RuntimeError("synth_hip_render failed with status %d" % rc)
    return out


def render_frames_to_host(syn, keys):
    """Both cameras of the frames `keys` of one generator, rendered on the device and copied back: uint8 array [2, len(keys), h, w].
    No torch involved (the parity test of the renderer runs in a process that holds the product library only)."""
    from .ctypes_types import RENDER_IMG
    w, h = syn.w, syn.h
    params = np.zeros(2 * len(keys), RENDER_IMG)
    for c in (0, 1):
        for i, k in enumerate(keys):
            params[c * len(keys) + i] = syn.render_params(k, c)
    r0, r1 = np.ascontiguousarray(syn.ray_table(0)), np.ascontiguousarray(syn.ray_table(1))
    out = np.empty((2, len(keys), h, w), np.uint8)
    rc = lib().synth_hip_render_host(params.ctypes.data, len(params), r0.ctypes.data, r1.ctypes.data, w, h, out.ctypes.data, w * h)
    if rc != 0:
        raise RuntimeError("synth_hip_render_host failed with status %d" % rc)
    return out
