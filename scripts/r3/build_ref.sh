#!/bin/bash
# builds build/variants/libfh_<name>.so from the sources of git ref $1 (A/B measurements on the GPU box)
set -e
REF=$1; NAME=$2; shift 2
D=/tmp/exp_$NAME; rm -rf $D; mkdir -p $D
git archive $REF faster_amd/csrc include | tar -x -C $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o build/variants/libfh_$NAME.so $D/faster_amd/csrc/fh_capi.hip $D/faster_amd/csrc/fh_pool.hip $D/faster_amd/csrc/fh_map.hip
