#!/bin/bash
# usage: sweep_scan.sh  -> prints scan kernel ms for several RLR_SCAN_VARIANT encodings
cd $GRAFT_REPO_ROOT
for v in 0x000004 0x000014 0x000002 0x000012 0x000008 0x000018 0x000404 0x001004 0x002004 0x000414 0x200004 0x100004 0x400804 0x000001; do
  RLR_SCAN_VARIANT=$v timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'scan_ms=%.4f'%d['roofline']['kernel_ms'], 'GBps=%.0f'%d['roofline']['achieved'], 'qps=%.1f'%d['value'])"
done
