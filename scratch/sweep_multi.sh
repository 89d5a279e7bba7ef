#!/bin/bash
# usage (via gpurun): bash scratch/sweep_multi.sh -- eight queries sharing one scan (scan_multi_kernel, 10 M x 768 f32) under
# RLR_SCAN_MULTI_VARIANT = workgroups per CU | rows per group << 8
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for w in 0 1 2 3 4 6; do for g in 8 32; do
  v=$(( w | (g << 8) )); [ $w = 0 ] && v=0; [ $w = 0 ] && [ $g != 8 ] && continue
  out=$(RLR_SCAN_MULTI_VARIANT=$v timeout -k 10 100 python3 $R/bench.py --batch 8 --steps 20 --warmup 3 --no-cpu --settle-ms 0 2>/dev/null | tail -n 1)
  echo "wgs=$w group=$g $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step %.4f qps %.1f" % (d["ms_per_step"], d["value"]))')"
done; done; done
