#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
CMD="python3 $R/bench.py --q8-scan --steps 30 --warmup 3 --no-cpu"
cd $R
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01e_q8_kt -- $CMD > $O/r01e_q8_kt.log 2>&1
grep '"metric"' $O/r01e_q8_kt.log | tail -1 > $O/r01e_bench_q8_scan_under_rocprof.json
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r01e_q8_fetch -- $CMD > $O/r01e_q8_fetch.log 2>&1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r01e_q8_write -- $CMD > $O/r01e_q8_write.log 2>&1
echo done
