#!/bin/bash
# one plain and one profiled run of the lexical timing script (the profiled one without the caller-thread sweep:
# concurrent submitters under rocprofv3's queue interception trip a ROCr / rocprofiler-sdk defect --
# profiles/r03_profiled_stress_aborts.md)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 200 python3 $R/scratch/time_lexical.py 200000 > $O/repro_lex_plain.log 2>&1 < /dev/null; echo "plain run rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/repro_lex_kt -- python3 $R/scratch/time_lexical.py 200000 --serial > $O/repro_lex_kt.log 2>&1 < /dev/null; echo "profiled run rc=$?"
