#!/bin/bash
# build/ru_one.hip with line tables -> /tmp/fh_code_size/{dev.elf,dis.txt}: the inputs of scripts/r6/isa_count.py and isa_of.py, in seconds
cd "$(dirname "$0")/../.."
T=/tmp/fh_code_size; mkdir -p $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm -gline-tables-only "$@" \
  -c --cuda-device-only -o $T/one.co build/ru_one.hip 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/one.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d -l --no-show-raw-insn $T/dev.elf > $T/dis.txt
