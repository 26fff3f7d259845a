#!/bin/bash
# Register / LDS / scratch use of every kernel of fh_capi.hip and fh_map.hip as the compiler reports it (same flags as faster_amd/build.py).
#   bash scripts/resource_usage.sh [extra flags] > profiles/rNN_kernel_resource_usage.txt
cd "$(dirname "$0")/.."
for src in faster_amd/csrc/fh_capi.hip faster_amd/csrc/fh_map.hip; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null $src 2>&1
done |
  grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {if (line) print line; line=$3; next} /^VGPRs:|^AGPRs:|Spill|ScratchSize|Occupancy|TotalSGPRs/ {line=line " | " $0} END {print line}' |
  c++filt | sed -e 's/(fh_problem const\*[^|]*|/ |/'
