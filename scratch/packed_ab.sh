#!/bin/bash
# A/B packed vs generic scan for every packed pitch: prints "dim dtype | packed GB/s | generic GB/s"
for spec in "64 f32" "128 f32" "320 f32" "384 f32" "640 f32" "896 f32" "1152 f32" "128 f16" "256 f16" "768 f16"; do
  set -- $spec
  bytes=$(( $1 * ( $2 == f16 ? 2 : 4 ) )); [ "$2" = f16 ] && bytes=$(( $1 * 2 )) || bytes=$(( $1 * 4 ))
  rows=$(( 24000000000 / bytes ))
  for v in 0 0x20; do
    RLR_SCAN_VARIANT=$v timeout -k 5 200 python bench.py --steps 30 --warmup 3 --dim $1 --dtype $2 --rows $rows --no-cpu --settle-ms 100 2>/dev/null | tail -1 > /tmp/_l$v.json
  done
  python -c "import json; a=json.load(open('/tmp/_l0.json')); b=json.load(open('/tmp/_l0x20.json')); print('$1 $2 rows=$rows | packed %.0f | generic %.0f' % (a['roofline']['achieved'], b['roofline']['achieved']))"
done
