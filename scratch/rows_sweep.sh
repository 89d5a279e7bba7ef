#!/bin/bash
# usage: rows_sweep.sh rows... ; prints rows, ms/step, kernel ms, GB/s  (env passes through)
for r in "$@"; do
  timeout -k 5 200 python bench.py --steps 200 --warmup 5 --rows $r --no-cpu 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print($r, round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved'],1))"
done
