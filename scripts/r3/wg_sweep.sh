#!/bin/bash
# C4 throughput against the number of resident solves per CU (driver configuration and one launch at a time)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for w in 0 10 9 8 7 6; do
  a=$(python bench.py --no-cpu --no-extra --wg-per-cu $w 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M pairs/s, %.2f ms/step' % (r['value']/1e6, r['ms_per_step']))")
  b=$(python bench.py --no-cpu --no-extra --wg-per-cu $w --inflight 1 --steps 24 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step alone' % r['ms_per_step'])")
  echo "workgroups per CU $w: $a; $b"
done
