#!/bin/bash
# pipelined C4 rate against fh_sched.waiting_workgroups: bash scripts/r6/waiting_sweep.sh
cd $GRAFT_REPO_ROOT
for w in 0 1 2 4 64; do
  python bench.py --no-cpu --no-extra --steps 96 --warmup 8 --waiting-workgroups $w 2>/dev/null > /tmp/ws.json
  python -c "import json; d=json.loads(open('/tmp/ws.json').read().strip().splitlines()[-1]); print('waiting $w', round(d['value']/1e6,2), 'M pairs/s')"
done
