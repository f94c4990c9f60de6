// host_math.h — tiny 3x3 double algebra for the host side of the C-ABI layer and the host mirror
// classes.  The summation order of every product is fixed ((a0*b0 + a1*b1) + a2*b2, starting from
// 0 + a0*b0) because some of these matrices (R_cam0_cam1, the essential matrix, K R K^-1) feed the
// bit-exact point arithmetic on the device; compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>

namespace hm {

struct Vec3 {
    double v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(double a, double b, double c) : v{a, b, c} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator-(const Vec3 &a) { return Vec3(-a[0], -a[1], -a[2]); }
inline Vec3 operator*(double s, const Vec3 &a) { return Vec3(s * a[0], s * a[1], s * a[2]); }
inline Vec3 operator*(const Vec3 &a, double s) { return Vec3(s * a[0], s * a[1], s * a[2]); }
inline Vec3 operator/(const Vec3 &a, double s) { return Vec3(a[0] / s, a[1] / s, a[2] / s); }
inline double dot(const Vec3 &a, const Vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const Vec3 &a) { return std::sqrt(dot(a, a)); }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

struct Mat3 {
    double m[9];
    Mat3() { std::memset(m, 0, sizeof(m)); }
    static Mat3 identity() { Mat3 r; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
    double &operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
    Mat3 transpose() const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i); return r; }
};
inline Mat3 operator*(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += a(i, k) * b(k, j);
        r(i, j) = s;
    }
    return r;
}
inline Vec3 operator*(const Mat3 &a, const Vec3 &b) {
    Vec3 r;
    for (int i = 0; i < 3; ++i) r[i] = a(i, 0) * b[0] + a(i, 1) * b[1] + a(i, 2) * b[2];
    return r;
}
inline Mat3 operator*(double s, const Mat3 &a) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i]; return r; }
inline Mat3 operator+(const Mat3 &a, const Mat3 &b) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
inline Mat3 operator-(const Mat3 &a, const Mat3 &b) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
inline Mat3 operator-(const Mat3 &a) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = -a.m[i]; return r; }
inline Mat3 skew(const Vec3 &w) {
    Mat3 r;
    r(0, 1) = -w[2]; r(0, 2) = w[1];
    r(1, 0) = w[2];  r(1, 2) = -w[0];
    r(2, 0) = -w[1]; r(2, 1) = w[0];
    return r;
}

struct Rigid {  // [R t; 0 1]
    Mat3 R; Vec3 t;
    Rigid() : R(Mat3::identity()) {}
    Rigid(const Mat3 &R_, const Vec3 &t_) : R(R_), t(t_) {}
    static Rigid from_rowmajor16(const double *a) {
        Rigid T;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T.R(i, j) = a[4 * i + j]; T.t[i] = a[4 * i + 3]; }
        return T;
    }
    Rigid inverse() const { Mat3 Rt = R.transpose(); return Rigid(Rt, -(Rt * t)); }
};
inline Rigid operator*(const Rigid &a, const Rigid &b) { return Rigid(a.R * b.R, a.R * b.t + a.t); }

}  // namespace hm
