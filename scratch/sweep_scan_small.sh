#!/bin/bash
# usage (via gpurun): bash scratch/sweep_scan_small.sh -- the single-query scan at small corpora (768-d f32) under RLR_SCAN_VARIANT:
# rows per wave step (bits 0-3), workgroups per CU (bits 8-15), rows per group (bits 16-23); scan stage time from the library's events
R=$GRAFT_REPO_ROOT
for n in 20000 50000 100000 200000 400000 1000000; do
 for bpc in 0 1 2 4 8; do for g in 4 8 16; do
  v=$(( 4 | (bpc << 8) | (g << 16) )); [ $bpc = 0 ] && v=0
  [ $bpc = 0 ] && [ $g != 4 ] && continue
  out=$(RLR_SCAN_VARIANT=$v timeout -k 10 60 python3 $R/scratch/time_c2_abi.py 100 $n 2>/dev/null | head -n 1)
  echo "n=$n bpc=$bpc group=$g $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("scan_us %.1f call_ms %.4f" % (d["scan"]*1000, d["abi_call_ms_profiled"]))')"
 done; done
done
