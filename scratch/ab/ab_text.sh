#!/bin/bash
# usage (via gpurun): bash scratch/ab/ab_text.sh -- two builds of the library (scratch/ab/librlr_gpu_{old,new}.so, made by hand) on one box,
# interleaved: config 2 with the query text at the C ABI (time_c2_text_abi.py)
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for which in new old; do
  cp $R/scratch/ab/librlr_gpu_$which.so $R/rust-local-rag_amd/librlr_gpu.so
  echo "$which $(timeout -k 10 200 python3 $R/scratch/time_c2_text_abi.py 2>/dev/null | tail -n 1)"
done; done
cp $R/scratch/ab/librlr_gpu_new.so $R/rust-local-rag_amd/librlr_gpu.so
