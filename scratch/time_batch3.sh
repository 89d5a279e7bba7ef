#!/bin/bash
timeout -k 5 300 python bench.py --batch 256 --image --steps 3 --warmup 1 --no-cpu --settle-ms 0 "$@" 2>/dev/null | tail -1 > /tmp/_b.json
python -c "import json; d=json.load(open('/tmp/_b.json')); print(round(d['ms_per_step'],3), d.get('stages_ms'))"
