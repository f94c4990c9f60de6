// system.h — host mirror of cg::System (reference msckf_core/include/system.h:16-41, src/system.cpp:11-54):
// one ImageProcessor + one MsckfVio sharing one device stream, three relay callbacks.
#pragma once
#include <string>
#include "image_processor.h"
#include "msckf_vio.h"

namespace cg {

class System {
  public:
    // reference constructor: camchain YAML path, app_*.yaml read from ../config (Q16); GPU 0
    explicit System(std::string file_cam_imu);
    // explicit configuration; ctx == nullptr creates a private context on `device`
    System(const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, mskf_ctx *ctx = nullptr, int device = 0);
    ~System();

    void stereo_callback(const cg::Image &cam0_img, const cg::Image &cam1_img, bool is_draw = false);
    void imu_callback(const cg::ImuConstPtr &msg);
    void backend_callback();

    std::vector<cg::Vector3> path_to_draw_;
    std::vector<cg::Point3f> points3d_to_draw_;
    cg::ImageProcessorPtr imgproc_ptr_;

    typedef std::shared_ptr<System> Ptr;
    typedef std::shared_ptr<const System> ConstPtr;

    // access for BatchRunner / tests
    cg::MsckfVioPtr msckfvio_ptr() const { return msckfvio_ptr_; }
    std::shared_ptr<CameraMeasurement> feature_msg() const { return feature_msg_ptr_; }
    void set_feature_msg(const std::shared_ptr<CameraMeasurement> &m) { feature_msg_ptr_ = m; }
    mskf_stream *stream() const { return stream_; }
    bool ok() const { return ok_; }
    bool copy_draw_buffers = true;   // backend_callback copies path_/points3d_ every frame in the reference (Q20)

  private:
    void setup(const mskf_calib &calib, const mskf_fe_cfg &fe, const mskf_ekf_cfg &ekf, mskf_ctx *ctx, int device);
    YAML::Node cfg_cam_imu_;
    std::shared_ptr<CameraMeasurement> feature_msg_ptr_;
    cg::MsckfVioPtr msckfvio_ptr_;
    mskf_ctx *own_ctx_ = nullptr;
    mskf_stream *stream_ = nullptr;
    bool ok_ = false;
};

typedef System::Ptr SystemPtr;
typedef System::ConstPtr SystemConstPtr;

}  // namespace cg
