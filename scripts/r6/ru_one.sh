#!/bin/bash
# Register / scratch use of ONE or three instantiations of the solve kernel (seconds instead of the 90 s of scripts/resource_usage.sh):
#   bash scripts/r6/ru_one.sh [-DRU_MORE] [-DRU_N=15] [other flags]
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null build/ru_one.hip 2>&1 |
  grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/ {if (line) print line; line=$3; next} /^VGPRs:|Spill|ScratchSize|Occupancy/ {line=line " | " $0} END {print line}' |
  c++filt | sed -e 's/(fh_problem const\*[^|]*|/ |/' | grep -v "^write_safe\|dt_initial_exact"
