#!/bin/bash
# build.sh <out name> [extra flags]   (run from anywhere)
D=$(cd "$(dirname "$0")" && pwd)
OUT=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I $D/../../include -Wno-unused-function "$@" -o $D/$OUT $D/gram_bench.hip 2>&1 | grep -E "error|Error" 
exit 0
# variants are produced by editing csrc/exact.hip (or stashing to get HEAD) and building under another name, e.g.
#   scratch/gram_bench/build.sh gb_cur; git stash; scratch/gram_bench/build.sh gb_head; git stash pop
#   gpurun -- 'bash scratch/gram_bench/run.sh gb_head; bash scratch/gram_bench/run.sh gb_cur'
