#!/bin/bash
# Builds oracle/_ref/libref_frontend.so from the reference's own front-end sources WHERE THEY LIE under /root/reference (never
# copied), plus the test-only shims of the libraries they name that are absent here (oracle/ref_frontend/shim/).  Outputs go only
# into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).  Exit 77: /root/reference is not there (GPU box) —
# the prebuilt library is used.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
REF=${FASTER_REFERENCE:-/root/reference}
OUT=$ROOT/oracle/_ref
if [ ! -d "$REF/thirdparty/jps3d/src/jps_planner" ] || [ ! -d "$REF/thirdparty/DecompROS/DecompUtil/include" ]; then
  echo "SKIPPED - $REF is not present: oracle/_ref/libref_frontend.so is used as built where it was" >&2
  exit 77
fi
mkdir -p "$OUT"
g++ -O2 -std=c++14 -fPIC -shared -w \
  -I "$HERE/shim" \
  -I "$REF/thirdparty/DecompROS/DecompUtil/include" \
  -I "$REF/thirdparty/jps3d/include" \
  "$REF/thirdparty/jps3d/src/jps_planner/graph_search.cpp" \
  "$REF/thirdparty/jps3d/src/jps_planner/jps_planner.cpp" \
  "$HERE/ref_frontend.cpp" \
  -o "$OUT/libref_frontend.so"
echo "built $OUT/libref_frontend.so"
