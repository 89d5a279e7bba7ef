#!/bin/bash
# The thread stress of the lexical index / fused text search under rocprofv3's queue interception -- ONE run per
# configuration (profiles/r03_profiled_stress_aborts.md has the cause of the round-2 aborts: ROCr 7.2's InterceptQueue hands
# rocprofiler-sdk's packet interceptor a run of packets that is not split at the ring's wrap-around once several host
# threads ring the doorbell of one shared hardware queue).
#   prof_stress.sh own-queues   every HIP stream gets its own hardware queue (GPU_MAX_HW_QUEUES=64): one producer per
#                               intercepted queue, runs of one packet -- the configuration the analysis predicts to be safe
#   prof_stress.sh shared       the runtime's default (4 hardware queues shared by all streams): the round-2 configuration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mode=${1:-shared}
cd /tmp && export TMPDIR=/tmp
if [ "$mode" = own-queues ]; then export GPU_MAX_HW_QUEUES=64; fi
rm -rf $O/stress_kt_$mode
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/stress_kt_$mode -- python3 $R/scratch/stress_lexical_threads.py 2 200 > $O/stress_r03_$mode.log 2>&1 < /dev/null
rc=$?
echo "profiled stress ($mode) rc=$rc"
grep -m1 "SIGSEGV\|stress ok\|AQL" $O/stress_r03_$mode.log
rm -rf $O/stress_kt_$mode   # the trace itself is not wanted, only whether the run survives
exit 0
