#!/bin/bash
# usage (via gpurun): bash scratch/prof_shard.sh [tag]  -- the shard-of-8 step: wall time unprofiled, then the kernel timeline under rocprofv3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-shard}
cd /tmp && export TMPDIR=/tmp
gcc -O2 -I $R/include $R/scratch/c_step.c -L $R/rust-local-rag_amd -lrlr_gpu -lm -Wl,-rpath,$R/rust-local-rag_amd -o /tmp/c_step && {
  for w in hybrid spin block; do echo -n "RLR_WAIT=$w  "; RLR_WAIT=$w timeout -k 10 60 /tmp/c_step 1250000 600; done
  for w in hybrid spin block; do echo -n "RLR_WAIT=$w  "; RLR_WAIT=$w timeout -k 10 60 /tmp/c_step 10000000 100; done
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_ckt -- /tmp/c_step 1250000 300 > $O/${TAG}_ckt.log 2>&1 < /dev/null
  f=$(ls $O/${TAG}_ckt/*/*kernel_trace.csv | tail -n 1); python3 $R/scratch/step_timeline.py $f | tee $O/${TAG}_c_timeline.txt; rm -rf $O/${TAG}_ckt
}
for m in plain sharded; do
  RLR_SHARD_MODE=$m timeout -k 10 120 python3 $R/scratch/time_shard_step.py 2>/dev/null | tail -n 1
done
RLR_SHARD_MODE=sharded timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_shard_step.py > $O/${TAG}_kt.log 2>&1 < /dev/null || { echo "profiled run failed"; tail -5 $O/${TAG}_kt.log; exit 1; }
f=$(ls $O/${TAG}_kt/*/*kernel_trace.csv | tail -n 1)
python3 $R/scratch/step_timeline.py $f | tee $O/${TAG}_timeline.txt
rm -rf $O/${TAG}_kt
