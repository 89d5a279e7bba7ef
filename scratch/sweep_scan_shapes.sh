#!/bin/bash
# usage (via gpurun): bash scratch/sweep_scan_shapes.sh -- one workgroup per CU (one wave per SIMD) against the round-1/2 launch shapes of
# the single-query scan, over row lengths and dtypes at 15-25 GB corpora (RLR_SCAN_VARIANT: rows per step | workgroups per CU << 8 |
# rows per group << 16; 0 = the library's own choice)
R=$GRAFT_REPO_ROOT
run() { # label, variant, rows, dim, dtype
  out=$(RLR_SCAN_VARIANT=$2 timeout -k 10 100 python3 $R/bench.py --steps 30 --warmup 4 --no-cpu --no-extras --rows $3 --dim $4 --dtype $5 --check 0 2>/dev/null | tail -n 1)
  echo "$5 dim=$4 rows=$3 $1 $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%s ms %.4f GBps %.0f" % (r["kernel"].split("<")[0], r["kernel_ms"], r["achieved"]))')"
}
shape() { # rows dim dtype
  run "default      " 0 $1 $2 $3
  run "r=4 bpc=1 g=8" $(( 4 | (1 << 8) | (8 << 16) )) $1 $2 $3
  run "r=2 bpc=1 g=8" $(( 2 | (1 << 8) | (8 << 16) )) $1 $2 $3
  run "r=4 bpc=1 g=16" $(( 4 | (1 << 8) | (16 << 16) )) $1 $2 $3
  run "r=4 bpc=2 g=8" $(( 4 | (2 << 8) | (8 << 16) )) $1 $2 $3
}
shape 24000000 256 f32
shape 12000000 512 f32
shape 7500000 1024 f32
shape 5000000 1536 f32
shape 3750000 2048 f32
shape 16000000 384 f32
shape 6000000 1000 f32
shape 24000000 512 f16
shape 12000000 1024 f16
shape 16000000 768 f16
shape 8000000 1536 f16
shape 6000000 2048 f16
shape 40000000 128 f32
