// oracle/o_math.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into the product.
//
// Small fixed-size linear algebra + JPL quaternion kinematics used by the CPU restatement of
// the reference's ImageProcessor / MsckfVio path.  The reference gets these from the absent
// vikit_cg (maths/, kinematics/); conventions follow SURVEY.md Appendix C, i.e. upstream
// MSCKF_VIO math_utils, whose names survive in the reference (msckf_vio.cpp:560 "skewSymmetric").
// parity unpinned: the reference ships no tests/fixtures for any of these.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <cassert>
#include <algorithm>
#include <array>

namespace orc {

struct V3 {
    double v[3];
    V3() : v{0, 0, 0} {}
    V3(double a, double b, double c) : v{a, b, c} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator-(const V3 &a) { return V3(-a[0], -a[1], -a[2]); }
inline V3 operator*(double s, const V3 &a) { return V3(s * a[0], s * a[1], s * a[2]); }
inline V3 operator*(const V3 &a, double s) { return V3(s * a[0], s * a[1], s * a[2]); }
inline V3 operator/(const V3 &a, double s) { return V3(a[0] / s, a[1] / s, a[2] / s); }
inline double dot(const V3 &a, const V3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 cross(const V3 &a, const V3 &b) {
    return V3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

struct M3 {
    double m[9];  // row-major
    M3() { std::memset(m, 0, sizeof(m)); }
    static M3 eye() { M3 r; r.m[0] = r.m[4] = r.m[8] = 1; return r; }
    double &operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
    M3 t() const { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i); return r; }
};
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 3; ++k) s += a(i, k) * b(k, j);
        r(i, j) = s;
    }
    return r;
}
inline V3 operator*(const M3 &a, const V3 &b) {
    V3 r; for (int i = 0; i < 3; ++i) r[i] = a(i, 0) * b[0] + a(i, 1) * b[1] + a(i, 2) * b[2];
    return r;
}
inline M3 operator*(double s, const M3 &a) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i]; return r; }
inline M3 operator+(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
inline M3 operator-(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
inline M3 operator-(const M3 &a) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = -a.m[i]; return r; }

// [w]x, SURVEY Appendix C
inline M3 skew(const V3 &w) {
    M3 r;
    r(0, 1) = -w[2]; r(0, 2) = w[1];
    r(1, 0) = w[2];  r(1, 2) = -w[0];
    r(2, 0) = -w[1]; r(2, 1) = w[0];
    return r;
}

// JPL quaternion [x y z w]; R(q) maps world -> body.
struct Quat {
    double q[4];
    Quat() : q{0, 0, 0, 1} {}
    Quat(double x, double y, double z, double w) : q{x, y, z, w} {}
    double &operator[](int i) { return q[i]; }
    double operator[](int i) const { return q[i]; }
};
inline Quat qnormalized(const Quat &a) {
    double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]);
    return Quat(a[0] / n, a[1] / n, a[2] / n, a[3] / n);
}
// R(q) = (2w^2-1) I - 2w [q_v]x + 2 q_v q_v^T
inline M3 quat_to_rot(const Quat &q) {
    V3 qv(q[0], q[1], q[2]);
    double w = q[3];
    M3 R = (2 * w * w - 1) * M3::eye() - (2 * w) * skew(qv);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) += 2 * qv[i] * qv[j];
    return R;
}
// upstream rotationToQuaternion (JPL)
inline Quat rot_to_quat(const M3 &R) {
    double tr = R(0, 0) + R(1, 1) + R(2, 2);
    double score[4] = {R(0, 0), R(1, 1), R(2, 2), tr};
    int mx = 0;
    for (int i = 1; i < 4; ++i) if (score[i] > score[mx]) mx = i;
    Quat q;
    if (mx == 0) {
        q[0] = std::sqrt(1 + 2 * R(0, 0) - tr) / 2.0;
        q[1] = (R(0, 1) + R(1, 0)) / (4 * q[0]);
        q[2] = (R(0, 2) + R(2, 0)) / (4 * q[0]);
        q[3] = (R(1, 2) - R(2, 1)) / (4 * q[0]);
    } else if (mx == 1) {
        q[1] = std::sqrt(1 + 2 * R(1, 1) - tr) / 2.0;
        q[0] = (R(0, 1) + R(1, 0)) / (4 * q[1]);
        q[2] = (R(1, 2) + R(2, 1)) / (4 * q[1]);
        q[3] = (R(2, 0) - R(0, 2)) / (4 * q[1]);
    } else if (mx == 2) {
        q[2] = std::sqrt(1 + 2 * R(2, 2) - tr) / 2.0;
        q[0] = (R(0, 2) + R(2, 0)) / (4 * q[2]);
        q[1] = (R(1, 2) + R(2, 1)) / (4 * q[2]);
        q[3] = (R(0, 1) - R(1, 0)) / (4 * q[2]);
    } else {
        q[3] = std::sqrt(1 + tr) / 2.0;
        q[0] = (R(1, 2) - R(2, 1)) / (4 * q[3]);
        q[1] = (R(2, 0) - R(0, 2)) / (4 * q[3]);
        q[2] = (R(0, 1) - R(1, 0)) / (4 * q[3]);
    }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    return qnormalized(q);
}
// q1 (x) q2 = L(q1) q2, normalised
inline Quat qmul(const Quat &a, const Quat &b) {
    Quat r;
    r[0] = a[3] * b[0] + a[2] * b[1] - a[1] * b[2] + a[0] * b[3];
    r[1] = -a[2] * b[0] + a[3] * b[1] + a[0] * b[2] + a[1] * b[3];
    r[2] = a[1] * b[0] - a[0] * b[1] + a[3] * b[2] + a[2] * b[3];
    r[3] = -a[0] * b[0] - a[1] * b[1] - a[2] * b[2] + a[3] * b[3];
    return qnormalized(r);
}
inline Quat small_angle_quat(const V3 &dtheta) {
    V3 a = dtheta / 2.0;
    double n = dot(a, a);
    Quat q;
    if (n <= 1) {
        q = Quat(a[0], a[1], a[2], std::sqrt(1 - n));
    } else {
        double s = std::sqrt(1 + n);
        q = Quat(a[0] / s, a[1] / s, a[2] / s, 1 / s);
    }
    return q;
}
// Hamilton quaternion (x y z w) of R, Eigen Quaterniond(R)
inline void rot_to_hamilton(const M3 &R, double out[4]) {
    double t = R(0, 0) + R(1, 1) + R(2, 2);
    double x, y, z, w;
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        w = 0.5 * t;
        t = 0.5 / t;
        x = (R(2, 1) - R(1, 2)) * t;
        y = (R(0, 2) - R(2, 0)) * t;
        z = (R(1, 0) - R(0, 1)) * t;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
        double c[3];
        c[i] = 0.5 * t;
        t = 0.5 / t;
        w = (R(k, j) - R(j, k)) * t;
        c[j] = (R(j, i) + R(i, j)) * t;
        c[k] = (R(k, i) + R(i, k)) * t;
        x = c[0]; y = c[1]; z = c[2];
    }
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}
// rotation angle of R, Eigen AngleAxisd(R).angle()
inline double rot_angle(const M3 &R) {
    double h[4];
    rot_to_hamilton(R, h);
    double n = std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    return 2.0 * std::atan2(n, std::fabs(h[3]));
}
// Eigen Quaterniond::FromTwoVectors(a, b).toRotationMatrix()
inline M3 from_two_vectors(const V3 &a, const V3 &b) {
    V3 v0 = a / norm(a), v1 = b / norm(b);
    double c = dot(v1, v0);
    double x, y, z, w;
    if (c < -1.0 + 1e-12) {
        // antiparallel: any axis orthogonal to v0 (not reached by the gravity initialisation)
        V3 ax = std::fabs(v0[0]) < 0.9 ? cross(v0, V3(1, 0, 0)) : cross(v0, V3(0, 1, 0));
        ax = ax / norm(ax);
        x = ax[0]; y = ax[1]; z = ax[2]; w = 0;
    } else {
        V3 axis = cross(v0, v1);
        double s = std::sqrt((1 + c) * 2);
        double invs = 1 / s;
        x = axis[0] * invs; y = axis[1] * invs; z = axis[2] * invs; w = s * 0.5;
    }
    // Hamilton quaternion -> rotation matrix
    M3 R;
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
    return R;
}
// OpenCV Rodrigues: rotation vector -> matrix
inline M3 rodrigues(const V3 &r) {
    double th = norm(r);
    if (th < 2.220446049250313e-16) return M3::eye();
    V3 k = r / th;
    double c = std::cos(th), s = std::sin(th), c1 = 1 - c;
    M3 R = c * M3::eye() + s * skew(k);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) += c1 * k[i] * k[j];
    return R;
}

// rigid transform; "T" = [R t; 0 1]
struct SE3 {
    M3 R; V3 t;
    SE3() : R(M3::eye()) {}
    SE3(const M3 &R_, const V3 &t_) : R(R_), t(t_) {}
    static SE3 from16(const double *a) {
        SE3 T;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T.R(i, j) = a[4 * i + j]; T.t[i] = a[4 * i + 3]; }
        return T;
    }
    SE3 inv() const { M3 Rt = R.t(); return SE3(Rt, -(Rt * t)); }
};
inline SE3 operator*(const SE3 &a, const SE3 &b) { return SE3(a.R * b.R, a.R * b.t + a.t); }

// dense row-major dynamic matrix
struct Mat {
    int r = 0, c = 0;
    std::vector<double> d;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    static Mat eye(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1; return m; }
    Mat t() const { Mat m(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(j, i) = (*this)(i, j); return m; }
    void set(int r0, int c0, const M3 &b) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) (*this)(r0 + i, c0 + j) = b(i, j); }
    void set(int r0, int c0, const Mat &b) { for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) (*this)(r0 + i, c0 + j) = b(i, j); }
    Mat block(int r0, int c0, int nr, int nc) const {
        Mat m(nr, nc);
        for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) m(i, j) = (*this)(r0 + i, c0 + j);
        return m;
    }
    void conservative_resize(int nr, int nc) {
        Mat m(nr, nc);
        for (int i = 0; i < std::min(r, nr); ++i) for (int j = 0; j < std::min(c, nc); ++j) m(i, j) = (*this)(i, j);
        *this = m;
    }
};
inline Mat operator*(const Mat &a, const Mat &b) {
    assert(a.c == b.r);
    Mat m(a.r, b.c);
    for (int i = 0; i < a.r; ++i)
        for (int k = 0; k < a.c; ++k) {
            double aik = a(i, k);
            if (aik == 0.0) continue;
            const double *bk = &b.d[(size_t)k * b.c];
            double *mi = &m.d[(size_t)i * m.c];
            for (int j = 0; j < b.c; ++j) mi[j] += aik * bk[j];
        }
    return m;
}
inline Mat operator+(const Mat &a, const Mat &b) { Mat m = a; for (size_t i = 0; i < m.d.size(); ++i) m.d[i] += b.d[i]; return m; }
inline Mat operator-(const Mat &a, const Mat &b) { Mat m = a; for (size_t i = 0; i < m.d.size(); ++i) m.d[i] -= b.d[i]; return m; }
inline Mat operator*(double s, const Mat &a) { Mat m = a; for (auto &x : m.d) x *= s; return m; }
inline Mat to_mat(const M3 &a) { Mat m(3, 3); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = a(i, j); return m; }

// In-place Cholesky solve of S X = B (S SPD n x n, B n x k), returns false if not PD.
// The reference uses Eigen LDLT (msckf_vio.cpp:850,924); for SPD S both give S^-1 B.
inline bool chol_solve(Mat S, Mat &B) {
    int n = S.r, k = B.c;
    for (int j = 0; j < n; ++j) {
        double s = S(j, j);
        for (int p = 0; p < j; ++p) s -= S(j, p) * S(j, p);
        if (!(s > 0)) return false;
        double l = std::sqrt(s);
        S(j, j) = l;
        for (int i = j + 1; i < n; ++i) {
            double t = S(i, j);
            for (int p = 0; p < j; ++p) t -= S(i, p) * S(j, p);
            S(i, j) = t / l;
        }
    }
    for (int c = 0; c < k; ++c) {
        for (int i = 0; i < n; ++i) {
            double t = B(i, c);
            for (int p = 0; p < i; ++p) t -= S(i, p) * B(p, c);
            B(i, c) = t / S(i, i);
        }
        for (int i = n - 1; i >= 0; --i) {
            double t = B(i, c);
            for (int p = i + 1; p < n; ++p) t -= S(p, i) * B(p, c);
            B(i, c) = t / S(i, i);
        }
    }
    return true;
}

// Deterministic atan / tan for the equidistant (fisheye) camera model.  cv::fisheye::distortPoints needs atan(r),
// cv::fisheye::undistortPoints needs tan(theta); libm and the GPU math library do not round them identically, so both
// sides evaluate the SAME sequence of IEEE double operations (+ - * / only, no contraction): argument reduction to a
// small interval and an odd Taylor polynomial.  Accuracy ~2e-16 relative (checked against libm in the tests).

// atan(x), x >= 0:  x > 1 -> pi/2 - atan(1/x);  then atan(x) = atan(c) + atan((x - c) / (1 + x c)) with the nearest
// c in {0, 1/4, 1/2, 3/4, 1} (|z| <= 1/8 -> 11 odd terms leave < 1e-21)
static inline double det_atan(double x) {
    const bool inv = x > 1.0;
    if (inv) x = 1.0 / x;
    double c = 0.0, ac = 0.0;
    if (x > 0.875)      { c = 1.0;  ac = 0.78539816339744830962; }
    else if (x > 0.625) { c = 0.75; ac = 0.64350110879328438680; }
    else if (x > 0.375) { c = 0.5;  ac = 0.46364760900080611621; }
    else if (x > 0.125) { c = 0.25; ac = 0.24497866312686415417; }
    const double z = (x - c) / (1.0 + x * c);
    const double z2 = z * z;
    double p = 1.0 / 21.0;
    p = 1.0 / 19.0 - z2 * p;
    p = 1.0 / 17.0 - z2 * p;
    p = 1.0 / 15.0 - z2 * p;
    p = 1.0 / 13.0 - z2 * p;
    p = 1.0 / 11.0 - z2 * p;
    p = 1.0 / 9.0 - z2 * p;
    p = 1.0 / 7.0 - z2 * p;
    p = 1.0 / 5.0 - z2 * p;
    p = 1.0 / 3.0 - z2 * p;
    p = 1.0 - z2 * p;
    const double a = ac + z * p;
    return inv ? 1.5707963267948966192 - a : a;
}
// tan(t), 0 <= t <= pi/2:  t > pi/4 -> 1 / tan(pi/2 - t);  tan = sin / cos with Taylor series on [0, pi/4]
static inline double det_tan(double t) {
    const bool inv = t > 0.78539816339744830962;
    if (inv) t = 1.5707963267948966192 - t;
    const double t2 = t * t;
    double s = 1.0 / 121645100408832000.0;              // 1/19!
    s = 1.0 / 355687428096000.0 - t2 * s;                // 1/17!
    s = 1.0 / 1307674368000.0 - t2 * s;                  // 1/15!
    s = 1.0 / 6227020800.0 - t2 * s;                     // 1/13!
    s = 1.0 / 39916800.0 - t2 * s;                       // 1/11!
    s = 1.0 / 362880.0 - t2 * s;                         // 1/9!
    s = 1.0 / 5040.0 - t2 * s;                           // 1/7!
    s = 1.0 / 120.0 - t2 * s;                            // 1/5!
    s = 1.0 / 6.0 - t2 * s;                              // 1/3!
    s = t * (1.0 - t2 * s);
    double c = 1.0 / 6402373705728000.0;                 // 1/18!
    c = 1.0 / 20922789888000.0 - t2 * c;                 // 1/16!
    c = 1.0 / 87178291200.0 - t2 * c;                    // 1/14!
    c = 1.0 / 479001600.0 - t2 * c;                      // 1/12!
    c = 1.0 / 3628800.0 - t2 * c;                        // 1/10!
    c = 1.0 / 40320.0 - t2 * c;                          // 1/8!
    c = 1.0 / 720.0 - t2 * c;                            // 1/6!
    c = 1.0 / 24.0 - t2 * c;                             // 1/4!
    c = 0.5 - t2 * c;                                    // 1/2!
    c = 1.0 - t2 * c;
    return inv ? c / s : s / c;
}

}  // namespace orc
