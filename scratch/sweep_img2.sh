#!/bin/bash
# usage (via gpurun): bash scratch/sweep_img2.sh -- the opt-in nomination scans (binary16 image: RLR_SCAN_IMAGE_VARIANT = chunks in flight |
# workgroups per CU << 8; 8-bit copy: RLR_Q8_VARIANT = rows in flight | workgroups per CU << 8 | group rows << 16) with few workgroups
# per CU, three interleaved repeats
R=$GRAFT_REPO_ROOT
img() { out=$(RLR_SCAN_IMAGE_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-extras --image-scan 2>/dev/null | tail -n 1)
  echo "image $1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms %.4f GBps %.0f qps %.1f" % (r["kernel_ms"], r["achieved"], d["value"]))')"; }
q8() { out=$(RLR_Q8_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-extras --q8-scan 2>/dev/null | tail -n 1)
  echo "q8 $1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms %.4f GBps %.0f qps %.1f" % (r["kernel_ms"], r["achieved"], d["value"]))')"; }
for rep in 1 2 3; do
  img "default(3 chunks x 4 wgs)" 0
  img "4 chunks x 1 wg" $(( 4 | (1 << 8) ))
  img "3 chunks x 2 wgs" $(( 3 | (2 << 8) ))
  img "3 chunks x 1 wg" $(( 3 | (1 << 8) ))
  q8 "default(4 wgs)" 0
  q8 "1 wg" $(( 1 << 8 ))
  q8 "2 wgs" $(( 2 << 8 ))
  q8 "3 wgs" $(( 3 << 8 ))
done
