#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for i in 1 2 3; do timeout -k 10 200 python3 $R/scratch/time_lexical.py 200000 > $O/repro_lex_$i.log 2>&1 < /dev/null; echo "plain run $i rc=$?"; done
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/repro_lex_kt$i -- python3 $R/scratch/time_lexical.py 200000 > $O/repro_lex_kt$i.log 2>&1 < /dev/null; echo "profiled run $i rc=$?"; grep -c "caller threads" $O/repro_lex_kt$i.log; done
