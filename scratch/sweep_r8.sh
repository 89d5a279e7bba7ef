#!/bin/bash
for rows in 10000000 5000000 2500000 1250000 300000; do
for v in 0 0x8 0x408 0x100408 0x200408 0x100008; do
  RLR_SCAN_VARIANT=$v timeout -k 5 200 python bench.py --steps 100 --warmup 10 --rows $rows --no-cpu --settle-ms 300 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print($rows, '$v', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_ms']*1e3,1), round(d['roofline']['achieved']))"
done; done
