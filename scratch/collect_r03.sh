#!/bin/bash
# usage (via gpurun): bash scratch/collect_r03.sh <tag>   -- ONCE, at the end of the round
# Every rocprofv3 pass behind the round's numbers, raw output under gpurun_out/<tag>_*; profiles/summarize.py condenses
# them afterwards.  Counter passes (--pmc) run on their own, never together with a trace domain.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, then the rocprofv3 arguments and the command
  local name=$1; shift
  timeout -k 10 280 rocprofv3 "$@" > $O/${TAG}_${name}.log 2>&1 < /dev/null || { echo "pass $name failed"; tail -3 $O/${TAG}_${name}.log; }
  echo "pass $name done"
}
HEAD="python3 $R/bench.py --steps 30 --warmup 3 --no-cpu --no-extras"
B256="python3 $R/bench.py --batch 256 --image --steps 12 --warmup 2 --no-cpu --settle-ms 0"
run head_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_head_kt -- $HEAD
grep -h '"metric"' $O/${TAG}_head_kt.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
run head_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_head_fetch -- $HEAD
run head_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_head_write -- $HEAD
run b256_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_b256_kt -- $B256
grep -h '"metric"' $O/${TAG}_b256_kt.log | tail -1 > $O/${TAG}_bench_batch256_image_under_rocprof.json
run b256_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_b256_sq1 -- $B256
run b256_sq2 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_b256_sq2 -- $B256
run b256_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_b256_fetch -- $B256
run b256_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_b256_write -- $B256
run c2_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_c2_kt -- python3 $R/scratch/time_c2_abi.py
run hyb_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_hyb_kt -- python3 $R/scratch/time_c2_hybrid.py hybrid-only
run c5_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_c5_kt -- python3 $R/scratch/time_c5_shard.py --image
run c5_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_c5_fetch -- python3 $R/scratch/time_c5_shard.py --image
run c5_write --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_c5_write -- python3 $R/scratch/time_c5_shard.py --image
run c5_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_c5_sq -- python3 $R/scratch/time_c5_shard.py --image
run mmr_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_mmr_kt -- python3 $R/scratch/time_mmr_f16.py 5
run mmr_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_mmr_sq -- python3 $R/scratch/time_mmr_f16.py 2
run mmr_fetch --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_mmr_fetch -- python3 $R/scratch/time_mmr_f16.py 2
run lex_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_lex_kt -- python3 $R/scratch/time_lexical.py 200000 --serial
run multi8_kt --kernel-trace --stats --output-format csv -d $O/${TAG}_multi8_kt -- python3 $R/bench.py --batch 8 --steps 20 --warmup 3 --no-cpu --settle-ms 0
grep -h '"metric"' $O/${TAG}_multi8_kt.log | tail -1 > $O/${TAG}_bench_batch8_under_rocprof.json
cd $R
timeout -k 10 300 python3 $R/scratch/time_multi_engine.py > $O/${TAG}_multi_engine_time.json 2> $O/${TAG}_multi_engine_time.err < /dev/null; echo "multi engine timing rc=$?"
timeout -k 10 200 python3 $R/bench.py --in-process --gpus 1 --steps 30 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_inprocess_n1.json
timeout -k 10 200 python3 $R/scratch/time_mmr_f16.py 5 > $O/${TAG}_mmr_f16_time.json 2>/dev/null
timeout -k 10 600 python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err < /dev/null
echo "default bench rc=$?"; head -c 300 $O/${TAG}_bench_n1.json; echo
ls $O | grep "^${TAG}_" | head -40
