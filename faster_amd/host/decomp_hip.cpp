// decomp_hip.cpp — see decomp_hip.hpp
#include "decomp_hip.hpp"

#include <cstdio>

DecompHip::~DecompHip() {
  if (ctx_) fh_destroy(ctx_);
}

bool DecompHip::ensureContext() {
  if (ctx_ && rc_ == FH_OK) return true;
  if (ctx_) return false;  // creation already failed: stay failed, loudly
  rc_ = fh_create(&ctx_, -1);
  if (rc_ != FH_OK) {
    err_ = ctx_ ? fh_last_error(ctx_) : "fh_create failed";
    std::fprintf(stderr, "DecompHip: device error: %s\n", err_.c_str());
    return false;
  }
  return true;
}

void DecompHip::setCloud(const std::vector<fhfront::V3>& cloud) {
  cloud_xyz_.resize(3 * cloud.size());
  for (size_t i = 0; i < cloud.size(); i++) {
    cloud_xyz_[3 * i] = cloud[i].x;
    cloud_xyz_[3 * i + 1] = cloud[i].y;
    cloud_xyz_[3 * i + 2] = cloud[i].z;
  }
}

std::vector<fhfront::LinearConstraint> DecompHip::cvxEllipsoidDecomp(const std::vector<fhfront::V3>& path, double drone_radius,
                                                                     double z_ground) {
  std::vector<fhfront::LinearConstraint> out;
  if (path.size() < 2) return out;
  if (!ensureContext()) return out;
  const int n_seg = (int)path.size() - 1;
  std::vector<double> segs(6 * (size_t)n_seg);
  for (int i = 0; i < n_seg; i++) {
    segs[6 * i + 0] = path[i].x; segs[6 * i + 1] = path[i].y; segs[6 * i + 2] = path[i].z;
    segs[6 * i + 3] = path[i + 1].x; segs[6 * i + 4] = path[i + 1].y; segs[6 * i + 5] = path[i + 1].z;
  }
  const double bbox[3] = {2.0, 2.0, 1.0};  // jps_manager.cpp:100
  const int max_faces = FH_MAX_FACES_POLY;
  std::vector<fh_face> faces((size_t)n_seg * max_faces);
  std::vector<int32_t> counts((size_t)n_seg);
  rc_ = fh_decompose_batch(ctx_, cloud_xyz_.empty() ? nullptr : cloud_xyz_.data(), (int)(cloud_xyz_.size() / 3), segs.data(), n_seg, bbox,
                           drone_radius, z_ground, max_faces, faces.data(), counts.data());
  if (rc_ != FH_OK) {
    err_ = fh_last_error(ctx_);
    std::fprintf(stderr, "DecompHip: device error: %s\n", err_.c_str());
    rc_ = FH_OK;  // the context stays usable; this call failed
    return out;
  }
  for (int i = 0; i < n_seg; i++) {
    if (counts[i] < 0) {  // more rows than setPolytopes accepts per polytope: report, never truncate silently
      err_ = "DecompHip: polytope " + std::to_string(i) + " needs more than FH_MAX_FACES_POLY rows";
      std::fprintf(stderr, "%s\n", err_.c_str());
      out.clear();
      return out;
    }
    fhfront::LinearConstraint c;
    for (int f = 0; f < counts[i]; f++) {
      const fh_face& r = faces[(size_t)i * max_faces + f];
      c.A.push_back(r.a[0]); c.A.push_back(r.a[1]); c.A.push_back(r.a[2]);
      c.b.push_back(r.b);
    }
    out.push_back(c);
  }
  return out;
}
