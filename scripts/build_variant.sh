#!/bin/bash
# A/B builds of the device library: bash scripts/build_variant.sh <name> [extra hipcc flags]  ->  build/libfasterhip_<name>.so
# (run with FASTERHIP_SO=build/libfasterhip_<name>.so; the product library faster_amd/libfasterhip.so is not touched)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm "$@" \
  -o build/libfasterhip_$name.so faster_amd/csrc/fh_capi.hip faster_amd/csrc/fh_pool.hip faster_amd/csrc/fh_map.hip
echo built build/libfasterhip_$name.so "$@"
