// Stub of <ros/package.h> for the ROS-free build of the UNTOUCHED reference solver (oracle/ref_harness/build.sh).
// faster/src/solverGurobi.cpp:13 includes it and uses nothing from it.
#pragma once
