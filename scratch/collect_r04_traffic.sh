#!/bin/bash
# usage (via gpurun): bash scratch/collect_r04_traffic.sh   -- counter traffic of the scan at the two per-GPU share shapes
# (1.25 M rows: the headline's share at 8 GPUs; 12.5 M rows: config 4's), separate --pmc passes; condensed by
# scratch/summarize_r04.sh into profiles/r04_shard1of8_pmc.json / r04_c4share_pmc.json, which bench.py quotes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r04
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 280 rocprofv3 "$@" > $O/${TAG}_${name}.log 2>&1 < /dev/null || { echo "pass $name failed"; tail -3 $O/${TAG}_${name}.log; }
  echo "pass $name done"; }
SH="python3 $R/scratch/time_shard_step.py"
C4="python3 $R/bench.py --rows 12500000 --steps 20 --warmup 3 --no-cpu --no-extras --settle-ms 500"
run shard_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_shard_fetch -- $SH
run shard_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_shard_write -- $SH
run c4_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_c4_kt -- $C4
run c4_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_c4_fetch -- $C4
run c4_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_c4_write -- $C4
