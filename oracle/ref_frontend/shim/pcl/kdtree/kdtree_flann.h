// TEST INFRASTRUCTURE (oracle/ref_frontend): the two PCL types MapUtil::readMap names (a point and a cloud of points); the
// kd-tree this header would bring is not used by the reference's map code.
#pragma once
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ {
  float x, y, z;
};
template <class P>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  std::vector<P> points;
};
}  // namespace pcl
