#!/bin/bash
for v in 0 0x400808 0x400408 0x401008 0x400810 0x400410 0x200810 0x400804 0x401004 0x200808 0x100808; do
  RLR_Q8_VARIANT=$v timeout -k 5 200 python bench.py --q8-scan --steps 100 --warmup 5 --no-cpu --settle-ms 200 2>/dev/null | tail -1 > /tmp/_l.json
  python -c "import json; d=json.load(open('/tmp/_l.json')); print('$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved'],1))"
done
