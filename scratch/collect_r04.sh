#!/bin/bash
# usage (via gpurun): bash scratch/collect_r04.sh [tag]   -- ONCE, at the end of the round, at the final tree
# Every rocprofv3 pass behind the round's numbers, raw output under gpurun_out/<tag>_*; scratch/summarize_r04.sh condenses
# them afterwards.  Counter passes (--pmc) run on their own, never together with a trace domain.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, then the rocprofv3 arguments and the command
  local name=$1; shift
  timeout -k 10 280 rocprofv3 "$@" > $O/${TAG}_${name}.log 2>&1 < /dev/null || { echo "pass $name failed"; tail -3 $O/${TAG}_${name}.log; }
  echo "pass $name done"
}
HEAD="python3 $R/bench.py --steps 30 --warmup 3 --no-cpu --no-extras"
B256="python3 $R/bench.py --batch 256 --image --steps 12 --warmup 2 --no-cpu --settle-ms 0"
B256P="python3 $R/bench.py --batch 256 --steps 8 --warmup 2 --no-cpu --settle-ms 0"
C5="python3 $R/scratch/time_c5_shard.py --image"
run head_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_head_kt -- $HEAD
grep -h '"metric"' $O/${TAG}_head_kt.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
run head_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_head_fetch -- $HEAD
run head_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_head_write -- $HEAD
run shard_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_shard_kt -- python3 $R/scratch/time_shard_step.py
f=$(ls $O/${TAG}_shard_kt/*/*kernel_trace.csv | tail -n 1); python3 $R/scratch/step_timeline.py $f > $O/${TAG}_shard_timeline.txt; tail -8 $O/${TAG}_shard_timeline.txt
run b256_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_b256_kt -- $B256
grep -h '"metric"' $O/${TAG}_b256_kt.log | tail -1 > $O/${TAG}_bench_batch256_image_under_rocprof.json
run b256_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_b256_fetch -- $B256
run b256_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_b256_write -- $B256
run b256p_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_b256p_kt -- $B256P
grep -h '"metric"' $O/${TAG}_b256p_kt.log | tail -1 > $O/${TAG}_bench_batch256_under_rocprof.json
run b256p_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_b256p_fetch -- $B256P
run b256p_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_b256p_write -- $B256P
run c2_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_c2_kt -- python3 $R/scratch/time_c2_abi.py
run hyb_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_hyb_kt -- python3 $R/scratch/time_c2_hybrid.py hybrid-only
run c5_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_c5_kt -- $C5
run c5_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_c5_fetch -- $C5
run c5_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_c5_write -- $C5
run c5_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_c5_sq -- $C5
run c5_sq2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_c5_sq2 -- $C5
run c5_sq3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC --output-format csv -d $O/${TAG}_c5_sq3 -- $C5
cd $R
timeout -k 10 200 python3 $R/bench.py --in-process --gpus 1 --steps 30 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_inprocess_n1.json
timeout -k 10 600 python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err < /dev/null
echo "default bench rc=$?"; head -c 300 $O/${TAG}_bench_n1.json; echo
ls $O | grep "^${TAG}_" | head -60
