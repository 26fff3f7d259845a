// DecompHip without a device: an empty corridor and a message, never a CPU decomposition (tests/test_host_class.py).
// With a device: prints the row count of each polytope.
#include <cstdio>

#include "decomp_hip.hpp"

int main() {
  DecompHip d;
  std::vector<fhfront::V3> cloud = {fhfront::V3(1.0, 0.6, 1.0), fhfront::V3(1.0, -0.7, 1.2), fhfront::V3(2.2, 0.5, 0.8)};
  std::vector<fhfront::V3> path = {fhfront::V3(0, 0, 1), fhfront::V3(1.5, 0, 1), fhfront::V3(3, 0.2, 1.2)};
  d.setCloud(cloud);
  const std::vector<fhfront::LinearConstraint> c = d.cvxEllipsoidDecomp(path, 0.05, 0.0);
  std::printf("{\"polytopes\": %zu, \"rows\": [", c.size());
  for (size_t i = 0; i < c.size(); i++) std::printf("%s%zu", i ? ", " : "", c[i].faces());
  std::printf("], \"error\": %d}\n", d.lastError().empty() ? 0 : 1);
  return 0;
}
