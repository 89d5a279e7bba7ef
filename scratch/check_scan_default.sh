#!/bin/bash
# usage (via gpurun): bash scratch/check_scan_default.sh -- the library's own scan launch shape against the round-2 shape forced through
# RLR_SCAN_VARIANT, at the shapes plan_scan now treats specially
R=$GRAFT_REPO_ROOT
run() { # label, variant, rows, dim, dtype
  out=$(RLR_SCAN_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 30 --warmup 4 --no-cpu --no-extras --rows $3 --dim $4 --dtype $5 2>/dev/null | tail -n 1)
  echo "$5 dim=$4 rows=$3 $1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms %.4f GBps %.0f qps %.1f" % (r["kernel_ms"], r["achieved"], d["value"]))')"
}
OLD=$(( 8 | (4 << 8) | (16 << 16) )); OLDS=$(( 4 | (8 << 8) | (16 << 16) ))
for n in 10000000 1000000; do run "new" 0 $n 768 f32; run "old" $OLD $n 768 f32; done
for n in 100000 30000; do run "new" 0 $n 768 f32; run "old(small)" $OLDS $n 768 f32; done
for n in 5000000 300000; do run "new" 0 $n 1536 f32; run "old" $OLD $n 1536 f32; done
for n in 8000000 300000; do run "new" 0 $n 1536 f16; run "old" $OLD $n 1536 f16; done
