// decomp_hip.hpp — the convex decomposition of FASTER's replan on the device, behind the call the reference makes.
//
// Reference: JPS_Manager::cvxEllipsoidDecomp(path, type_space, l_constraints, poly_out)
// (/root/reference/faster/src/jps_manager.cpp:80-133), called twice per replan (faster.cpp:398, :499).  DecompHip is the
// same step through the C ABI (fh_decompose_batch, include/fasterhip.h): one wavefront per path segment on the GPU; the
// result is the std::vector of constraints `setPolytopes` takes.  No CPU fallback: without a device the call returns an
// empty corridor and lastError() says why (the caller's replan then fails, as a failed decomposition would).
#pragma once
#include <string>
#include <vector>

#include "../../include/fasterhip.h"
#include "corridor_frontend.hpp"

class DecompHip {
public:
  DecompHip() = default;
  ~DecompHip();
  DecompHip(const DecompHip&) = delete;
  DecompHip& operator=(const DecompHip&) = delete;

  // obstacle cloud (the map's occupied / unknown+occupied points of jps_manager.cpp:88-98); kept until the next call
  void setCloud(const std::vector<fhfront::V3>& cloud);
  // one polytope per path segment: separating planes, the local bounding box (2,2,1) and the ground plane -z <= -z_ground
  std::vector<fhfront::LinearConstraint> cvxEllipsoidDecomp(const std::vector<fhfront::V3>& path, double drone_radius, double z_ground);
  // the call shape of fhfront::decompose_path (used as the decomposition policy of fhreplan::Planner)
  std::vector<fhfront::LinearConstraint> operator()(const std::vector<fhfront::V3>& path, const std::vector<fhfront::V3>& cloud,
                                                    double drone_radius, double z_ground) {
    setCloud(cloud);
    return cvxEllipsoidDecomp(path, drone_radius, z_ground);
  }
  const std::string& lastError() const { return err_; }
  int deviceStatus() const { return rc_; }

private:
  bool ensureContext();
  fh_ctx* ctx_ = nullptr;
  int rc_ = FH_OK;
  std::string err_;
  std::vector<double> cloud_xyz_;
};

// the host policy: DecompUtil's algorithm on the CPU (faster_amd/host/corridor_frontend.hpp)
struct DecompHost {
  std::vector<fhfront::LinearConstraint> operator()(const std::vector<fhfront::V3>& path, const std::vector<fhfront::V3>& cloud,
                                                    double drone_radius, double z_ground) const {
    return fhfront::decompose_path(path, cloud, drone_radius, z_ground);
  }
};
