#!/bin/bash
# gpurun --timeout 600 -- 'bash scratch/ab_gemm8.sh "0 1 2 4 6"'   same-box A/B of the gemm8 variants (two rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for round in 1 2; do
for v in $1; do
  RLR_GEMM8_VARIANT=$v timeout -k 5 200 python bench.py --batch 256 --image --steps 30 --warmup 3 --no-cpu --settle-ms 300 2>/dev/null | tail -1 > gpurun_out/ab_$v.json || exit 1
  python -c "import json; d=json.load(open('gpurun_out/ab_$v.json')); print('variant $v:', round(d['ms_per_step'],3), round(d['value']), d.get('stages_ms'))"
done; done
