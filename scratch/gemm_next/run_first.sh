#!/bin/bash
# First GPU call of the next round (via gpurun, from the repo root):
#   gpurun --timeout 300 -- 'bash scratch/gemm_next/run_first.sh'
# Builds the library-layout 8-phase harness against the in-tree librlr_gpu.so and runs its three checks
# (materialised scores, filter-mode candidate sets, one-process A/B timing at 10 M x 768 x 256).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/include -o /tmp/gemm8_lib $R/scratch/gemm_next/gemm8_lib.hip \
    -L $R/rust-local-rag_amd -lrlr_gpu -Wl,-rpath,$R/rust-local-rag_amd
timeout -k 5 120 /tmp/gemm8_lib > $R/gpurun_out/gemm8_lib.log 2>&1 || { echo "gemm8_lib failed"; tail -20 $R/gpurun_out/gemm8_lib.log; exit 1; }
tail -20 $R/gpurun_out/gemm8_lib.log
