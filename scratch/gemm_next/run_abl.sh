#!/bin/bash
# gpurun --timeout 300 -- 'bash scratch/gemm_next/run_abl.sh'
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm8_abl $R/scratch/gemm_next/gemm8_abl.hip
timeout -k 5 120 /tmp/gemm8_abl > $R/gpurun_out/gemm8_abl.log 2>&1 || { echo "gemm8_abl failed"; tail -20 $R/gpurun_out/gemm8_abl.log; exit 1; }
cat $R/gpurun_out/gemm8_abl.log
