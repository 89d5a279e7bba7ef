#!/bin/bash
for v in 0x803 0x403 0x1003 0x802 0x402 0x804 0x404 0x1004 0x806 0x406 0x206; do
  RLR_SCAN_IMAGE_VARIANT=$v timeout -k 5 200 python bench.py --image-scan --steps 60 --warmup 5 --no-cpu --settle-ms 200 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print('$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved'],1))"
done
