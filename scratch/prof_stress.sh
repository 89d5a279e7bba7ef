#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  rm -rf $O/stress_kt
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/stress_kt -- python3 $R/scratch/stress_lexical_threads.py 2 200 > $O/stress_kt_$i.log 2>&1 < /dev/null; echo "profiled stress $i rc=$?"
  grep -c "errors 0" $O/stress_kt_$i.log; grep -m1 "SIGSEGV\|stress ok" $O/stress_kt_$i.log
done
