#!/bin/bash
# Register / LDS / scratch use of every kernel of fh_capi.hip as the compiler reports it (same flags as faster_amd/build.py).
#   bash scripts/resource_usage.sh [extra flags] > profiles/rNN_kernel_resource_usage.txt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null faster_amd/csrc/fh_capi.hip 2>&1 |
  grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {if (line) print line; line=$3; next} /^VGPRs:|^AGPRs:|Spill|ScratchSize|Occupancy|TotalSGPRs/ {line=line " | " $0} END {print line}' |
  c++filt | sed -e 's/(fh_problem const\*[^|]*|/ |/'
