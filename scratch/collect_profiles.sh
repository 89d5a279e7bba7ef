#!/bin/bash
# usage (on the GPU box, via gpurun): scratch/collect_profiles.sh <tag>
# rocprofv3 kernel trace + two PMC passes of the default bench, the batched (image) bench trace, and an
# unprofiled default bench run; raw output under gpurun_out/, summaries are made afterwards by profiles/summarize.py
set -e
TAG=${1:-r01d}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
CMD="python3 $R/bench.py --steps 30 --warmup 3 --no-cpu"
cd $R
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- $CMD > $O/${TAG}_kt.log 2>&1
grep '"metric"' $O/${TAG}_kt.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
echo "kernel trace done"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $CMD > $O/${TAG}_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $CMD > $O/${TAG}_write.log 2>&1
echo "write pass done"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_batch -- python3 $R/bench.py --batch 256 --image --steps 10 --warmup 2 --no-cpu > $O/${TAG}_kt_batch.log 2>&1
grep '"metric"' $O/${TAG}_kt_batch.log | tail -1 > $O/${TAG}_bench_batch256_image_under_rocprof.json
echo "batch trace done"
timeout -k 10 400 python3 $R/bench.py > $O/${TAG}_bench_n1.log 2>&1
grep '"metric"' $O/${TAG}_bench_n1.log | tail -1 > $O/${TAG}_bench_n1.json
cat $O/${TAG}_bench_n1.json
