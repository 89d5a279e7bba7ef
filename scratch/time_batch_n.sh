#!/bin/bash
# usage: time_batch_n.sh <batch> [extra args]
b=$1; shift
timeout -k 5 300 python bench.py --batch $b --image --steps 20 --warmup 3 --no-cpu --settle-ms 0 "$@" 2>/dev/null | tail -1 > /tmp/_b.json
python -c "import json; d=json.load(open('/tmp/_b.json')); print('batch $b', round(d['ms_per_step'],3), round(d['value']), d.get('stages_ms'))"
