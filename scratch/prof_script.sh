#!/bin/bash
# gpurun --timeout 600 -- 'bash scratch/prof_script.sh <tag> <script.py> [args]'   rocprofv3 kernel trace + stats of a python script
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 "$@" > $O/${TAG}_kt.log 2>&1 < /dev/null
rc=$?
S=$(find $O/${TAG}_kt -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -z "$S" ]; then echo "no stats (rc=$rc)"; tail -5 $O/${TAG}_kt.log; exit 1; fi
cp "$S" $O/${TAG}_kernel_stats.csv
grep -h '^{' $O/${TAG}_kt.log | tail -3
cut -c1-170 < "$S" | sed -n 1,16p
