#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 600 python scripts/r3/diag2.py > $O/diag2_new.txt 2>&1; cat $O/diag2_new.txt
FASTERHIP_SO=$R/build/variants/libfh_lb3.so timeout 600 python scripts/r3/diag2.py > $O/diag2_old.txt 2>&1; cat $O/diag2_old.txt
