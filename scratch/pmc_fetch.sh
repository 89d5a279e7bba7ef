#!/bin/bash
# FETCH_SIZE (KiB, x2 on gfx950) of the batched image GEMM kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/bench.py --batch 256 --image --steps 3 --warmup 1 --no-cpu --settle-ms 0 > $R/gpurun_out/pmc_fetch.log 2>&1) || { tail -5 $R/gpurun_out/pmc_fetch.log; exit 1; }
python3 $R/scratch/pmc_dump.py gemm_ $R/gpurun_out/pmc_fetch
