// Stub of <decomp_ros_utils/data_ros_utils.h> for the ROS-free build of the UNTOUCHED reference solver.
// faster/include/solverGurobi.hpp:19 includes the ROS header only to reach LinearConstraint3D; the type itself is header-only
// DecompUtil code (thirdparty/DecompROS/DecompUtil/include/decomp_geometry/polyhedron.h:115-185), included from the reference
// tree where it lies (build.sh adds that include path).  Needs Eigen3 (EIGEN3_INCLUDE_DIR).
#pragma once
#include <decomp_geometry/polyhedron.h>
