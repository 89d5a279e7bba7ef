#!/bin/bash
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_lex_kt
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_lex_kt -- python3 $R/scratch/time_lexical.py 200000 --serial > $O/${TAG}_lex_kt.log 2>&1 < /dev/null; echo "lex pass rc=$?"
tail -3 $O/${TAG}_lex_kt.log
