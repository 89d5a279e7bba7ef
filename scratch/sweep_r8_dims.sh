#!/bin/bash
for spec in "256 f32 24000000" "512 f32 12000000" "1024 f32 6000000" "768 f16 16000000" "1024 f16 12000000" "512 f16 24000000"; do
set -- $spec
for v in 0 0x100408 0x408; do
  RLR_SCAN_VARIANT=$v timeout -k 5 200 python bench.py --steps 60 --warmup 5 --dim $1 --dtype $2 --rows $3 --no-cpu --settle-ms 300 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print('$1 $2', '$v', round(d['roofline']['kernel_ms']*1e3,1), round(d['roofline']['achieved']))"
done; done
