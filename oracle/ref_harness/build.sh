#!/bin/bash
# Builds the UNTOUCHED reference solver (/root/reference/faster/src/solverGurobi.cpp) behind ROS-free stubs into oracle/_ref/ref_driver,
# following the reference's own build lines (faster/CMakeLists.txt:8-12 Eigen3 + GUROBI include dirs, :54-55 libgurobi_c++.a + libgurobi*.so).
# TEST INFRASTRUCTURE: the Gurobi tie-breaker of SURVEY.md 8(c)(5).  Needs what this container does not have:
#   GUROBI_HOME          a Gurobi installation with a valid licence (8.1 / 9.0 / 9.1 were tested by the reference, Readme.md:43)
#   EIGEN3_INCLUDE_DIR   Eigen 3 headers (default /usr/include/eigen3)
#   REFERENCE            the reference tree (default /root/reference)
# Nothing is copied from the reference: its sources are compiled where they lie; outputs go to oracle/_ref/ only (git-ignored).
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
REF=${REFERENCE:-/root/reference}
EIGEN=${EIGEN3_INCLUDE_DIR:-/usr/include/eigen3}
if [ -z "${GUROBI_HOME:-}" ] || [ ! -f "$GUROBI_HOME/include/gurobi_c++.h" ]; then
  echo "oracle/ref_harness: SKIPPED — GUROBI_HOME is not set or has no include/gurobi_c++.h (Gurobi is closed source and absent here);"
  echo "  parity with the reference stays UNPINNED; the oracle is pinned as described in its header."
  exit 77
fi
if [ ! -f "$EIGEN/Eigen/Dense" ]; then
  echo "oracle/ref_harness: SKIPPED — Eigen3 headers not found at $EIGEN (set EIGEN3_INCLUDE_DIR)"; exit 77
fi
if [ ! -f "$REF/faster/src/solverGurobi.cpp" ]; then
  echo "oracle/ref_harness: SKIPPED — reference tree not found at $REF"; exit 77
fi
mkdir -p "$OUT"
GLIB=$(ls "$GUROBI_HOME"/lib/libgurobi[0-9]*.so 2>/dev/null | head -1)
set -x
g++ -O2 -std=c++11 -I "$HERE/stubs" -I "$REF/faster/include" -I "$REF/thirdparty/DecompROS/DecompUtil/include" -I "$EIGEN" -I "$GUROBI_HOME/include" \
    "$HERE/ref_driver.cpp" "$REF/faster/src/solverGurobi.cpp" -o "$OUT/ref_driver" \
    "$GUROBI_HOME/lib/libgurobi_c++.a" "$GLIB" -lpthread -lm -Wl,-rpath,"$GUROBI_HOME/lib"
