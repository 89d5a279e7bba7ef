#!/bin/bash
# gpurun --timeout 600 -- 'bash scratch/prof_batch.sh <tag> [bench args]'   kernel trace of the batched (image) bench
TAG=${1:-r02_batch256}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/bench.py --batch 256 --image --steps 20 --warmup 3 --no-cpu "$@" > $O/${TAG}_kt.log 2>&1 || { tail -5 $O/${TAG}_kt.log; exit 1; }
grep '"metric"' $O/${TAG}_kt.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
S=$(find $O/${TAG}_kt -name '*kernel_stats.csv' | head -1)
cp $S $O/${TAG}_kernel_stats.csv
head -12 $S | cut -c1-200
