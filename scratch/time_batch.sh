#!/bin/bash
# usage: time_batch.sh [extra bench args]; prints ms/step, q/s, gemm ms for the 256-query image batch
timeout -k 5 300 python bench.py --batch 256 --image --steps 20 --warmup 3 --no-cpu --settle-ms 0 "$@" 2>/dev/null | tail -1 > /tmp/_b.json
python -c "import json; d=json.load(open('/tmp/_b.json')); print(round(d['ms_per_step'],3), round(d['value']), d.get('stages_ms'))"
