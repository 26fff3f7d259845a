// faster_stub.hpp — ROS/Eigen/PCL-free stand-ins for the two types that cross the solver boundary.
//
// The reference solver header drags ROS in through decomp_ros_utils (faster/include/solverGurobi.hpp:19) only to
// reach `LinearConstraint3D`, and Eigen for `state` (faster/include/faster_types.hpp:79-165).  Neither ROS nor
// Eigen exists in this image, so the host class is built against these minimal equivalents, which expose exactly
// the members the solver path touches:
//   state:              pos, vel, accel, jerk (3-vectors with x()/y()/z()), yaw, dyaw, setPos/setVel/setAccel/
//                       setJerk/setZero                                   (faster_types.hpp:79-149)
//   LinearConstraint3D: A() (F x 3), b() (F), inside(pt)                  (DecompUtil polyhedron.h:115-185)
// When FASTER is built with its real dependencies, define FASTER_HIP_USE_REFERENCE_TYPES: this header then
// includes the reference's own headers — faster_types.hpp (which expects Eigen and the standard containers to be included before it)
// and DecompUtil's polyhedron.h — instead of defining the stand-ins, so that solver_hip.cpp compiles as a translation unit of its
// own against FASTER's types (tests/test_host_class.py::test_reference_types_build_round_trip does exactly that).
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>
#ifdef FASTER_HIP_USE_REFERENCE_TYPES
#include <Eigen/Dense>
#include <iostream>
#include <string>
#include <decomp_geometry/polyhedron.h>
#include "faster_types.hpp"
#endif

namespace fhstub {

struct Vec3 {
  double v[3];
  Vec3() : v{0, 0, 0} {}
  Vec3(double x_, double y_, double z_) : v{x_, y_, z_} {}
  static Vec3 Zero() { return Vec3(); }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
  double dot(const Vec3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  Vec3 operator+(const Vec3& o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vec3 operator-(const Vec3& o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vec3 operator*(double s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
  const Vec3& transpose() const { return *this; }
  void setZero() { v[0] = v[1] = v[2] = 0; }
};

// F x 3 row-major matrix with the accessors the solver needs.
struct MatX3 {
  std::vector<double> d;
  MatX3() {}
  explicit MatX3(std::size_t rows) : d(rows * 3, 0.0) {}
  std::size_t rows() const { return d.size() / 3; }
  std::size_t cols() const { return 3; }
  double& operator()(std::size_t r, std::size_t c) { return d[r * 3 + c]; }
  double operator()(std::size_t r, std::size_t c) const { return d[r * 3 + c]; }
};

struct VecX {
  std::vector<double> d;
  VecX() {}
  explicit VecX(std::size_t n) : d(n, 0.0) {}
  std::size_t rows() const { return d.size(); }
  std::size_t size() const { return d.size(); }
  double& operator()(std::size_t i) { return d[i]; }
  double operator()(std::size_t i) const { return d[i]; }
  double& operator[](std::size_t i) { return d[i]; }
  double operator[](std::size_t i) const { return d[i]; }
};

}  // namespace fhstub

#ifndef FASTER_HIP_USE_REFERENCE_TYPES

struct state {
  fhstub::Vec3 pos, vel, accel, jerk;
  double yaw = 0;
  double dyaw = 0;
  void setPos(double x, double y, double z) { pos = fhstub::Vec3(x, y, z); }
  void setVel(double x, double y, double z) { vel = fhstub::Vec3(x, y, z); }
  void setAccel(double x, double y, double z) { accel = fhstub::Vec3(x, y, z); }
  void setJerk(double x, double y, double z) { jerk = fhstub::Vec3(x, y, z); }
  void setPos(const fhstub::Vec3& p) { pos = p; }
  void setVel(const fhstub::Vec3& p) { vel = p; }
  void setAccel(const fhstub::Vec3& p) { accel = p; }
  void setJerk(const fhstub::Vec3& p) { jerk = p; }
  void setYaw(double y) { yaw = y; }
  void setZero() {
    pos.setZero(); vel.setZero(); accel.setZero(); jerk.setZero();
    yaw = 0; dyaw = 0;
  }
};

// A x <= b, one polytope of the corridor.
struct LinearConstraint3D {
  LinearConstraint3D() {}
  LinearConstraint3D(const fhstub::MatX3& A, const fhstub::VecX& b) : A_(A), b_(b) {}
  // strict rejection like the reference: a point ON a face is inside (polyhedron.h:155-164)
  bool inside(const fhstub::Vec3& pt) const {
    for (std::size_t i = 0; i < b_.rows(); i++)
      if (A_(i, 0) * pt.x() + A_(i, 1) * pt.y() + A_(i, 2) * pt.z() - b_(i) > 0) return false;
    return true;
  }
  fhstub::MatX3 A() const { return A_; }
  fhstub::VecX b() const { return b_; }
  fhstub::MatX3 A_;
  fhstub::VecX b_;
};

#endif  // FASTER_HIP_USE_REFERENCE_TYPES
