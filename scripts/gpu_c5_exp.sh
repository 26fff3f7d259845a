#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in 1 2 8; do
timeout 300 python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --steps 6 --warmup 2 --inflight $f > /tmp/o.json 2>/tmp/o.err; python -c "
import json; d=json.load(open('/tmp/o.json')); print('c5 inflight $f', round(d['value']), round(d['ms_per_step'],1))"
done
