#!/bin/bash
# usage (via gpurun): bash scratch/sweep_scan_10m.sh -- the headline scan (10 M x 768 f32) under RLR_SCAN_VARIANT: rows per wave step
# (bits 0-3), non-temporal loads off (bit 4), workgroups per CU (bits 8-15), rows per group (bits 16-23); the scan kernel's HIP-event
# time from the bench line; every candidate several times, interleaved (allocation placement moves a run by 1-3 %)
R=$GRAFT_REPO_ROOT
run() { # label, variant
  out=$(RLR_SCAN_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -n 1)
  echo "$1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("scan_ms %.4f frac %.4f qps %.1f" % (d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]))')"
}
for rep in 1 2 3; do
  run "library default (r=4 bpc=1 g=8)" 0
  run "r=3 bpc=1 group=6" $(( 3 | (1 << 8) | (6 << 16) ))
  run "r=3 bpc=1 group=9" $(( 3 | (1 << 8) | (9 << 16) ))
  run "r=5 bpc=1 group=10" $(( 5 | (1 << 8) | (10 << 16) ))
  run "r=6 bpc=1 group=6" $(( 6 | (1 << 8) | (6 << 16) ))
  run "r=6 bpc=1 group=12" $(( 6 | (1 << 8) | (12 << 16) ))
  run "r=3 bpc=2 group=6" $(( 3 | (2 << 8) | (6 << 16) ))
done
