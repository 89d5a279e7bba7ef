#!/bin/bash
# gpurun --timeout 900 -- 'bash scratch/run_gemm8_tests.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or image or mfma" > gpurun_out/g8_tests.log 2>&1
rc=$?; tail -5 gpurun_out/g8_tests.log
[ $rc -ne 0 ] && exit $rc
for g in 1 0; do
  echo "RLR_GEMM8=$g"
  RLR_GEMM8=$g timeout -k 5 300 python bench.py --batch 256 --image --steps 20 --warmup 3 --no-cpu --settle-ms 0 2>/dev/null | tail -1 > gpurun_out/g8_bench_$g.json || exit 1
  python -c "import json; d=json.load(open('gpurun_out/g8_bench_$g.json')); print(round(d['ms_per_step'],3), round(d['value']), d.get('stages_ms'), d['band_retries'])"
done
