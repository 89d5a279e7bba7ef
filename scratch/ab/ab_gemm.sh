#!/bin/bash
# usage (via gpurun): bash scratch/ab/ab_gemm.sh -- two builds of the library (scratch/ab/librlr_gpu_{old,new}.so, made by hand) on one
# box, interleaved: config 3 (bench.py --batch 256 --image) and the config 5 share (time_c5_shard.py --image)
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for which in new old; do
  cp $R/scratch/ab/librlr_gpu_$which.so $R/rust-local-rag_amd/librlr_gpu.so
  out=$(timeout -k 10 100 python3 $R/bench.py --batch 256 --image --steps 12 --warmup 2 --no-cpu --settle-ms 0 2>/dev/null | tail -n 1)
  echo "C3 $which $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_batch %.4f main_ms %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))')"
  out=$(timeout -k 10 200 python3 $R/scratch/time_c5_shard.py --image 2>/dev/null | tail -n 1)
  echo "C5 $which $(echo $out | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("search_ms %.3f gemm_ms %.3f" % (d["batched_search_top308_ms"], d["gemm_ms"]))')"
done; done
cp $R/scratch/ab/librlr_gpu_new.so $R/rust-local-rag_amd/librlr_gpu.so
