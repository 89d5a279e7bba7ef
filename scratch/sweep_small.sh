#!/bin/bash
# scan variants at a 1.25M-row shard: variant = R | blocks/CU << 8 | group << 16
for v in 0 0x100004 0x200004 0x400004 0x100002 0x200002 0x100408 0x200404 0x201004 0x100404 0x101004 0x200008; do
  RLR_SCAN_VARIANT=$v timeout -k 5 200 python bench.py --steps 400 --warmup 20 --rows 1250000 --no-cpu --settle-ms 300 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print('$v', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_ms']*1e3,1), round(d['roofline']['achieved']))"
done
