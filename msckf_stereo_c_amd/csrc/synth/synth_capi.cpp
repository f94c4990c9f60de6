// synth_capi.cpp — C entry points of the synthetic EuRoC-shaped generator for ctypes (tests, bench).
#include "synth.h"

extern "C" {

void synth_euroc_calib(int width, int height, mskf_calib *out) { *out = synth::euroc_calib(width, height); }

void *synth_create(uint32_t seed, int width, int height, int n_static, int n_loop, double pixel_sigma, double motion_scale) {
    synth::Cfg c;
    c.seed = seed; c.width = width; c.height = height;
    c.calib = synth::euroc_calib(width, height);
    c.n_static = n_static; c.n_loop = n_loop; c.pixel_sigma = pixel_sigma; c.motion_scale = motion_scale;
    return new synth::Stream(c);
}
void synth_destroy(void *h) { delete (synth::Stream *)h; }
void synth_render(void *h, int k, uint8_t *cam0, uint8_t *cam1) { ((synth::Stream *)h)->render(k, cam0, cam1); }
double synth_frame_time(void *h, int k) { return ((synth::Stream *)h)->frame_time(k); }
void synth_imu(void *h, int j, mskf_imu_sample *out) { *out = ((synth::Stream *)h)->imu_sample(j); }
void synth_gt_pose(void *h, int k, mskf_pose *out) { *out = ((synth::Stream *)h)->gt_pose(k); }
// inputs of the device renderer (synth_render.hip): per-image parameters and the per-camera ray table (w x h x 2 floats)
void synth_render_params(void *h, int k, int cam, synth::RenderImg *out) { *out = ((synth::Stream *)h)->render_params(k, cam); }
const float *synth_ray_table(void *h, int cam) { return ((synth::Stream *)h)->ray_table(cam); }
int synth_imu_per_frame(void *h) { return ((synth::Stream *)h)->cfg().imu_per_frame; }

}  // extern "C"
