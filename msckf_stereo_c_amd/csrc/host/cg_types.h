// cg_types.h — value types of namespace cg used at the reference's class boundary.
//
// The reference takes these from the absent vikit_cg (maths/vector.h, cv/yimg.h, cv/types.h) and
// from msckf_core/include/common/data_msg.h:15-55.  Only the members the boundary actually uses
// are provided (SURVEY.md §2.3 "vikit_cg API surface actually used"); the message structs keep the
// reference's field order and types so a FeatureMeasurement is the same 40-byte record.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#include "../abi/host_math.h"

namespace cg {

typedef double FLOAT;
typedef hm::Vec3 Vector3;
typedef hm::Mat3 Mat3;

struct Vector4 {
    double v[4];
    Vector4() : v{0, 0, 0, 0} {}
    Vector4(double a, double b, double c, double d) : v{a, b, c, d} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};

struct Point2f {
    float x, y;
    Point2f() : x(0.f), y(0.f) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Point3f {
    float x, y, z;
    Point3f() : x(0.f), y(0.f), z(0.f) {}
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct Size2 { int width, height; int area() const { return width * height; } };

// 8-bit gray image, dense row-major (cg::YImg8: rows(), cols(), data(), size().area(), empty())
class YImg8 {
  public:
    YImg8() : rows_(0), cols_(0) {}
    YImg8(int rows, int cols) : rows_(rows), cols_(cols), d_((size_t)rows * cols, 0) {}
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    uint8_t *data() { return d_.data(); }
    const uint8_t *data() const { return d_.data(); }
    Size2 size() const { return Size2{cols_, rows_}; }
    bool empty() const { return d_.empty(); }

  private:
    int rows_, cols_;
    std::vector<uint8_t> d_;
};

// data_msg.h:15-26
struct Image {
    double time_stamp;
    YImg8 image;
};
typedef std::shared_ptr<Image> ImagePtr;
typedef std::shared_ptr<const Image> ImageConstPtr;

struct Imu {
    double time_stamp;
    Vector3 angular_velocity;
    Vector3 linear_acceleration;
};
typedef std::shared_ptr<Imu> ImuPtr;
typedef std::shared_ptr<const Imu> ImuConstPtr;

// data_msg.h:30-46
struct FeatureMeasurement {
    unsigned int id;
    double u0;  // horizontal coordinate in cam0
    double v0;  // vertical coordinate in cam0
    double u1;  // horizontal coordinate in cam1
    double v1;  // vertical coordinate in cam1
};
static_assert(sizeof(FeatureMeasurement) == 40, "FeatureMeasurement must stay the reference's 40-byte record");
typedef std::shared_ptr<FeatureMeasurement> FeatureMeasurementPtr;
typedef std::shared_ptr<const FeatureMeasurement> FeatureMeasurementConstPtr;

struct CameraMeasurement {
    double time_stamp;
    std::vector<FeatureMeasurement> features;
};
typedef std::shared_ptr<CameraMeasurement> CameraMeasurementPtr;
typedef std::shared_ptr<const CameraMeasurement> CameraMeasurementConstPtr;

// data_msg.h:48-55
struct TrackingInfo {
    double time_stamp;
    int before_tracking;
    int after_tracking;
    int after_matching;
    int after_ransac;
};

}  // namespace cg
