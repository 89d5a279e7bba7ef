#!/bin/bash
# small batches on the default configuration (no nomination copies): shared scan vs the matrix-core pipeline
for b in 2 4 8; do
  for m in "" 1; do
    RLR_NO_MULTI_SCAN=$m timeout -k 5 300 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu --settle-ms 0 2>/dev/null | tail -1 > /tmp/_b.json
    python -c "import json; d=json.load(open('/tmp/_b.json')); print('batch $b no_multi=$m', round(d['ms_per_step'],3), round(d['value']), d.get('stages_ms'), 'fallbacks', d.get('band_retries'))"
  done
done
