#!/bin/bash
# The thread stress of the lexical index / fused text search under rocprofv3's queue interception -- ONE run per
# configuration (profiles/r03_profiled_stress_aborts.md has the cause of the round-2 aborts: ROCr 7.2's InterceptQueue hands
# rocprofiler-sdk's packet interceptor a run of packets that is not split at the ring's wrap-around once several host
# threads ring the doorbell of one shared hardware queue).
#   prof_stress.sh [own-queues] every HIP stream gets its own hardware queue (GPU_MAX_HW_QUEUES=64): one producer per
#                               intercepted queue, runs of one packet -- the safe configuration, and the default
#   prof_stress.sh KNOWN-CRASH-shared   the runtime's default (4 hardware queues shared by all streams): the round-2
#                               configuration, which ABORTS under the profiler (SIGSEGV in librocprofiler-sdk, or an invalid
#                               AQL packet and a ~4-minute wait for the outer timeout).  The cause is established; nothing
#                               needs this run again -- it is only reachable by spelling the argument out.
# Exits with the profiled run's return code.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mode=${1:-own-queues}
if [ "$mode" = KNOWN-CRASH-shared ]; then mode=shared; elif [ "$mode" != own-queues ]; then echo "usage: prof_stress.sh [own-queues | KNOWN-CRASH-shared]"; exit 2; fi
cd /tmp && export TMPDIR=/tmp
if [ "$mode" = own-queues ]; then export GPU_MAX_HW_QUEUES=64; fi
rm -rf $O/stress_kt_$mode
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/stress_kt_$mode -- python3 $R/scratch/stress_lexical_threads.py 2 200 > $O/stress_r03_$mode.log 2>&1 < /dev/null
rc=$?
echo "profiled stress ($mode) rc=$rc"
grep -m1 "SIGSEGV\|stress ok\|AQL" $O/stress_r03_$mode.log
rm -rf $O/stress_kt_$mode   # the trace itself is not wanted, only whether the run survives
exit $rc
