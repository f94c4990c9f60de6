// kinematics.h — JPL quaternion helpers for the host mirror of cg::MsckfVio.
// The reference takes these from the absent vikit_cg kinematics/ (Quarternion, RotationMatrix,
// from_two_vector, quarternion_hamilton); the conventions are those of upstream MSCKF_VIO
// math_utils: q = [x y z w], R(q) maps world -> body, q1 (x) q2 = L(q1) q2.
#pragma once
#include <cmath>
#include "../abi/host_math.h"

namespace kin {
using hm::Mat3;
using hm::Vec3;

inline void normalize4(double *q) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

inline Mat3 rotation_of(const double *q) {
    const Vec3 qv(q[0], q[1], q[2]);
    const double w = q[3];
    Mat3 R = (2 * w * w - 1) * Mat3::identity() - (2 * w) * hm::skew(qv);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) += 2 * qv[i] * qv[j];
    return R;
}

inline void quaternion_of(const Mat3 &R, double *q) {
    const double tr = R(0, 0) + R(1, 1) + R(2, 2);
    const double score[4] = {R(0, 0), R(1, 1), R(2, 2), tr};
    int best = 0;
    for (int i = 1; i < 4; ++i) if (score[i] > score[best]) best = i;
    switch (best) {
        case 0:
            q[0] = std::sqrt(1 + 2 * R(0, 0) - tr) / 2.0;
            q[1] = (R(0, 1) + R(1, 0)) / (4 * q[0]); q[2] = (R(0, 2) + R(2, 0)) / (4 * q[0]); q[3] = (R(1, 2) - R(2, 1)) / (4 * q[0]);
            break;
        case 1:
            q[1] = std::sqrt(1 + 2 * R(1, 1) - tr) / 2.0;
            q[0] = (R(0, 1) + R(1, 0)) / (4 * q[1]); q[2] = (R(1, 2) + R(2, 1)) / (4 * q[1]); q[3] = (R(2, 0) - R(0, 2)) / (4 * q[1]);
            break;
        case 2:
            q[2] = std::sqrt(1 + 2 * R(2, 2) - tr) / 2.0;
            q[0] = (R(0, 2) + R(2, 0)) / (4 * q[2]); q[1] = (R(1, 2) + R(2, 1)) / (4 * q[2]); q[3] = (R(0, 1) - R(1, 0)) / (4 * q[2]);
            break;
        default:
            q[3] = std::sqrt(1 + tr) / 2.0;
            q[0] = (R(1, 2) - R(2, 1)) / (4 * q[3]); q[1] = (R(2, 0) - R(0, 2)) / (4 * q[3]); q[2] = (R(0, 1) - R(1, 0)) / (4 * q[3]);
    }
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    normalize4(q);
}

// out = a (x) b, normalised
inline void multiply(const double *a, const double *b, double *out) {
    double r[4];
    r[0] = a[3] * b[0] + a[2] * b[1] - a[1] * b[2] + a[0] * b[3];
    r[1] = -a[2] * b[0] + a[3] * b[1] + a[0] * b[2] + a[1] * b[3];
    r[2] = a[1] * b[0] - a[0] * b[1] + a[3] * b[2] + a[2] * b[3];
    r[3] = -a[0] * b[0] - a[1] * b[1] - a[2] * b[2] + a[3] * b[3];
    normalize4(r);
    for (int i = 0; i < 4; ++i) out[i] = r[i];
}

inline void small_angle(const Vec3 &dtheta, double *q) {
    const Vec3 a = dtheta / 2.0;
    const double n = hm::dot(a, a);
    if (n <= 1) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = std::sqrt(1 - n); }
    else { const double s = std::sqrt(1 + n); q[0] = a[0] / s; q[1] = a[1] / s; q[2] = a[2] / s; q[3] = 1 / s; }
}

// Hamilton quaternion (x y z w) of a rotation matrix, as Eigen::Quaterniond(R) (msckf_vio.cpp:1251)
inline void hamilton_of(const Mat3 &R, double *out) {
    double t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        out[3] = 0.5 * t;
        t = 0.5 / t;
        out[0] = (R(2, 1) - R(1, 2)) * t; out[1] = (R(0, 2) - R(2, 0)) * t; out[2] = (R(1, 0) - R(0, 1)) * t;
        return;
    }
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    out[i] = 0.5 * t;
    t = 0.5 / t;
    out[3] = (R(k, j) - R(j, k)) * t;
    out[j] = (R(j, i) + R(i, j)) * t;
    out[k] = (R(k, i) + R(i, k)) * t;
}

// Eigen::AngleAxisd(R).angle() (msckf_vio.cpp:1054)
inline double angle_of(const Mat3 &R) {
    double h[4];
    hamilton_of(R, h);
    return 2.0 * std::atan2(std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]), std::fabs(h[3]));
}

// Eigen::Quaterniond::FromTwoVectors(a, b).toRotationMatrix() (msckf_vio.cpp:236)
inline Mat3 shortest_arc(const Vec3 &a, const Vec3 &b) {
    const Vec3 v0 = a / hm::norm(a), v1 = b / hm::norm(b);
    const double c = hm::dot(v1, v0);
    double x, y, z, w;
    if (c < -1.0 + 1e-12) {
        Vec3 ax = std::fabs(v0[0]) < 0.9 ? hm::cross(v0, Vec3(1, 0, 0)) : hm::cross(v0, Vec3(0, 1, 0));
        ax = ax / hm::norm(ax);
        x = ax[0]; y = ax[1]; z = ax[2]; w = 0;
    } else {
        const Vec3 axis = hm::cross(v0, v1);
        const double s = std::sqrt((1 + c) * 2), invs = 1 / s;
        x = axis[0] * invs; y = axis[1] * invs; z = axis[2] * invs; w = s * 0.5;
    }
    Mat3 R;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
    return R;
}

}  // namespace kin
