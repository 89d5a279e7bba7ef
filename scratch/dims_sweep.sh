#!/bin/bash
# usage: dims_sweep.sh "dim dtype rows" ... ; prints dim dtype rows ms/step kernel-ms GB/s
for spec in "$@"; do
  set -- $spec
  timeout -k 5 200 python bench.py --steps 60 --warmup 5 --dim $1 --dtype $2 --rows $3 --no-cpu --settle-ms 200 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print('$1 $2 $3', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved'],1), d['roofline']['kernel'])"
done
