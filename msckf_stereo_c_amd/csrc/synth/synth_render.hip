// synth_render.hip — the synthetic stereo sequences of bench.py rendered ON THE DEVICE.
//
// Test / bench input generator only (like synth.h): no part of the VIO pipeline and not part of the C-ABI of include/mskf_hip.h.
// Round 3 rendered the looping sequences of a rank on the host before anything was timed: 192 sequences x 85 stereo pairs took
// 52 s with 14 threads on a 16-core share and 11.8 GB of host memory per rank - start-up cost that scales with ranks x host
// cores on a node whose cores are shared by eight ranks.  A pixel of the generator is a pure function of (camera pose of the
// frame, the pixel's undistorted ray, the seed): one thread per pixel evaluates synth::shade_pixel, the SAME source the host
// renderer runs (synth.h, compiled with -ffp-contract=off on both sides: single IEEE operations, identical bytes; checked in
// tests/test_gpu_kernels.py).  The host still does what needs libm, once per image: the pose of the frame (sin / cos) and the
// per-camera ray tables (synth::Stream::render_params / ray_table).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include "synth.h"

__global__ __launch_bounds__(256) void k_synth_render(const synth::RenderImg *imgs, const float *rays0, const float *rays1, uint8_t *out, size_t frame_bytes, int n_px) {
    const synth::RenderImg I = imgs[blockIdx.y];
    const float *rays = I.cam ? rays1 : rays0;
    uint8_t *dst = out + (size_t)blockIdx.y * frame_bytes;
    // four neighbouring pixels per thread, stored as one dword
    const int base = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (base >= n_px) return;
    if (base + 3 < n_px) {
        const float4 r01 = *reinterpret_cast<const float4 *>(rays + 2 * (size_t)base), r23 = *reinterpret_cast<const float4 *>(rays + 2 * (size_t)base + 4);
        const uint32_t p0 = synth::shade_pixel(I, r01.x, r01.y, (uint32_t)base), p1 = synth::shade_pixel(I, r01.z, r01.w, (uint32_t)base + 1);
        const uint32_t p2 = synth::shade_pixel(I, r23.x, r23.y, (uint32_t)base + 2), p3 = synth::shade_pixel(I, r23.z, r23.w, (uint32_t)base + 3);
        *reinterpret_cast<uint32_t *>(dst + base) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
    } else {
        for (int i = base; i < n_px; ++i) dst[i] = synth::shade_pixel(I, rays[2 * (size_t)i], rays[2 * (size_t)i + 1], (uint32_t)i);
    }
}

extern "C" {
// Render n_imgs images of width x height pixels: image i goes to out_dev + i * frame_bytes (device memory, frame_bytes a multiple
// of 4), from imgs_host[i] (synth_render_params) and the two cameras' ray tables (device memory).  Everything is enqueued on
// `stream` (a hipStream_t; nullptr = the null stream) and waited for: a bench that gives every pipeline stage a hardware queue of
// its own passes one of those streams, so that the generator does not bind one more queue.  Returns 0 on success.
int synth_hip_render(const synth::RenderImg *imgs_host, int n_imgs, const float *rays0_dev, const float *rays1_dev, int width, int height,
                     uint8_t *out_dev, size_t frame_bytes, void *stream) {
    if (!imgs_host || n_imgs <= 0 || !rays0_dev || !rays1_dev || !out_dev || width <= 0 || height <= 0 || (frame_bytes & 3) || frame_bytes < (size_t)width * height) return -1;
    hipStream_t st = (hipStream_t)stream;
    // the records go through a pinned staging block of the library's own rather than straight from the caller's pageable memory (a
    // numpy array), which the runtime would have to pin in place for an asynchronous copy.  (Suspected for a while of the bench's
    // multi-second stalls in round 4; those were the SDMA staging copies of the pipeline itself)
    const size_t pbytes = sizeof(synth::RenderImg) * (size_t)n_imgs;
    synth::RenderImg *d = nullptr, *hp = nullptr;
    if (hipHostMalloc((void **)&hp, pbytes, hipHostMallocDefault) != hipSuccess) return -2;
    if (hipMalloc((void **)&d, pbytes) != hipSuccess) { (void)hipHostFree(hp); return -2; }
    std::memcpy(hp, imgs_host, pbytes);
    int rc = 0;
    if (hipMemcpyAsync(d, hp, pbytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = -2;
    const int n_px = width * height;
    for (int i0 = 0; rc == 0 && i0 < n_imgs; i0 += 32768) {          // (grid.y limit)
        const int cnt = n_imgs - i0 < 32768 ? n_imgs - i0 : 32768;
        hipLaunchKernelGGL(k_synth_render, dim3((n_px + 1023) / 1024, cnt), dim3(256), 0, st, d + i0, rays0_dev, rays1_dev, out_dev + (size_t)i0 * frame_bytes, frame_bytes, n_px);
        if (hipGetLastError() != hipSuccess) rc = -3;
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == 0) rc = -3;
    (void)hipFree(d);
    (void)hipHostFree(hp);
    return rc;
}

// The same from and to HOST memory (ray tables w x h x 2 floats each, out n_imgs x frame_bytes): device buffers are allocated
// and released inside.  For the parity test (no other GPU library in the process) and small renders.
int synth_hip_render_host(const synth::RenderImg *imgs_host, int n_imgs, const float *rays0_host, const float *rays1_host, int width, int height,
                          uint8_t *out_host, size_t frame_bytes) {
    if (!rays0_host || !rays1_host || !out_host || width <= 0 || height <= 0 || n_imgs <= 0) return -1;
    const size_t rb = sizeof(float) * 2 * (size_t)width * height, ob = frame_bytes * (size_t)n_imgs;
    float *r0 = nullptr, *r1 = nullptr; uint8_t *o = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&r0, rb) != hipSuccess || hipMalloc((void **)&r1, rb) != hipSuccess || hipMalloc((void **)&o, ob) != hipSuccess) rc = -2;
    if (rc == 0 && (hipMemcpy(r0, rays0_host, rb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(r1, rays1_host, rb, hipMemcpyHostToDevice) != hipSuccess)) rc = -2;
    if (rc == 0) rc = synth_hip_render(imgs_host, n_imgs, r0, r1, width, height, o, frame_bytes, nullptr);
    if (rc == 0 && hipMemcpy(out_host, o, ob, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
    if (r0) (void)hipFree(r0);
    if (r1) (void)hipFree(r1);
    if (o) (void)hipFree(o);
    return rc;
}
}
