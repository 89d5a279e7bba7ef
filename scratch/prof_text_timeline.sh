#!/bin/bash
# usage (via gpurun): bash scratch/prof_text_timeline.sh [tag] -- kernel timeline of the text search (both streams)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-txt}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/scratch/time_c2_hybrid.py hybrid-only > $O/${TAG}_time.log 2>&1 < /dev/null; tail -n 1 $O/${TAG}_time.log
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_c2_hybrid.py hybrid-only > $O/${TAG}_kt.log 2>&1 < /dev/null || { echo "profiled run failed"; exit 1; }
f=$(ls $O/${TAG}_kt/*/*kernel_trace.csv | tail -n 1)
python3 $R/scratch/text_timeline.py $f mmr_greedy > $O/${TAG}_timeline.txt; cat $O/${TAG}_timeline.txt
rm -rf $O/${TAG}_kt
