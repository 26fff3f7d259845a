// TEST INFRASTRUCTURE (oracle/ref_frontend): jps3d's map_util.h includes "ros/ros.h" but uses nothing of it.
#pragma once
