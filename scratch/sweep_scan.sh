#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0x000000 0x200000 0x100000 0x200002 0x201000 0x200400 0x000002 0x300000; do
  RLR_SCAN_VARIANT=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'scan_ms=%.4f'%d['roofline']['kernel_ms'], 'GBps=%.0f'%d['roofline']['achieved'], 'qps=%.1f'%d['value'])"
done
