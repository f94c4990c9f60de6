// system.cpp — see system.h.  Reference: msckf_core/src/system.cpp:11-54.
#include "system.h"
#include <iostream>

namespace cg {

System::System(std::string file_cam_imu) : feature_msg_ptr_(new CameraMeasurement) {
    try {
        cfg_cam_imu_ = YAML::LoadFile(file_cam_imu);
        mskf_calib calib = calib_from_yaml(cfg_cam_imu_);
        mskf_fe_cfg fe = fe_cfg_from_yaml(YAML::LoadFile("../config/app_imgproc.yaml"));
        mskf_ekf_cfg ekf = ekf_cfg_from_yaml(YAML::LoadFile("../config/app_msckfvio.yaml"));
        setup(calib, fe, ekf, nullptr, 0);
        if (ok_) { imgproc_ptr_->enableFileOutputs(); msckfvio_ptr_->enableFileOutputs(); }   // pose_out.txt, debug_imageprocessor.txt
    } catch (const std::exception &e) {   // the reference swallows init errors and only prints (system.cpp:17-33)
        std::cerr << "Cannot initialize System: " << e.what() << std::endl;
    }
}

System::System(const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, mskf_ctx *ctx, int device)
    : feature_msg_ptr_(new CameraMeasurement) {
    setup(calib, fe, ekf, ctx, device);
}

void System::setup(const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, mskf_ctx *ctx, int device) {
    if (!ctx) {
        int rc = mskf_ctx_create(device, &own_ctx_);
        if (rc != MSKF_OK) { std::cerr << "Cannot create the device context: " << mskf_last_error() << std::endl; return; }
        ctx = own_ctx_;
    }
    int rc = mskf_stream_create(ctx, &calib, &fe, &ekf, &stream_);
    if (rc != MSKF_OK) { std::cerr << "Cannot create the device stream: " << mskf_last_error() << std::endl; return; }
    imgproc_ptr_.reset(new cg::ImageProcessor(calib, fe));
    imgproc_ptr_->attach(stream_);
    if (!imgproc_ptr_->initialize()) { std::cerr << "Cannot initialize Image Processor..." << std::endl; return; }
    msckfvio_ptr_.reset(new cg::MsckfVio(calib, ekf));
    msckfvio_ptr_->attach(stream_);
    if (!msckfvio_ptr_->initialize()) { std::cerr << "Cannot initialize MsckfVio..." << std::endl; return; }
    ok_ = true;
}

System::~System() {
    imgproc_ptr_.reset();
    msckfvio_ptr_.reset();
    if (stream_) mskf_stream_destroy(stream_);
    if (own_ctx_) mskf_ctx_destroy(own_ctx_);
}

// system.cpp:40-43
void System::stereo_callback(const cg::Image &cam0_img, const cg::Image &cam1_img, bool is_draw) {
    if (!ok_) return;
    imgproc_ptr_->stereoCallback(cam0_img, cam1_img, is_draw);
    feature_msg_ptr_ = imgproc_ptr_->feature_msg_ptr_;
}

// system.cpp:45-48
void System::imu_callback(const cg::ImuConstPtr &msg) {
    if (!ok_) return;
    imgproc_ptr_->imuCallback(msg);
    msckfvio_ptr_->imuCallback(msg);
}

// system.cpp:50-54
void System::backend_callback() {
    if (!ok_) return;
    if (feature_msg_ptr_ == imgproc_ptr_->feature_msg_ptr_) msckfvio_ptr_->setZeroTailHint(feature_msg_ptr_.get(), imgproc_ptr_->zeroTailStart());
    msckfvio_ptr_->featureCallback(feature_msg_ptr_);
    if (copy_draw_buffers) {
        path_to_draw_ = msckfvio_ptr_->get_path();
        points3d_to_draw_ = msckfvio_ptr_->get_points3d();
    }
}

}  // namespace cg
