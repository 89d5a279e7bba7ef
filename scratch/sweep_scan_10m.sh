#!/bin/bash
# usage (via gpurun): bash scratch/sweep_scan_10m.sh -- the headline scan (10 M x 768 f32) under RLR_SCAN_VARIANT: rows per wave step
# (bits 0-3), non-temporal loads off (bit 4), software-pipelined kernel (bit 6), workgroups per CU (bits 8-15), rows per group
# (bits 16-23); the scan kernel's HIP-event time from the bench line; every candidate several times, interleaved (allocation
# placement moves a run by 1-3 %)
R=$GRAFT_REPO_ROOT
run() { # label, variant
  out=$(RLR_SCAN_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -n 1)
  echo "$1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("scan_ms %.4f frac %.4f qps %.1f" % (d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]))')"
}
for rep in 1 2 3; do
  run "library default (r=4 bpc=1 g=8)" 0
  for g in 8 16 32 64; do run "pipe r=4 bpc=1 group=$g" $(( 4 | 64 | (1 << 8) | (g << 16) )); done
  run "pipe r=2 bpc=1 group=8" $(( 2 | 64 | (1 << 8) | (8 << 16) ))
  run "pipe r=4 bpc=2 group=8" $(( 4 | 64 | (2 << 8) | (8 << 16) ))
  run "pipe r=2 bpc=2 group=8" $(( 2 | 64 | (2 << 8) | (8 << 16) ))
done
