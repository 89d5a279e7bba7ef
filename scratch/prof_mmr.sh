#!/bin/bash
# kernel trace + matrix-core counters of the batched MMR at config 5's pool shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03_mmr}
python3 $R/scratch/time_mmr_f16.py 5 > $O/${TAG}_plain.json 2> $O/${TAG}_plain.err; cat $O/${TAG}_plain.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_kt $O/${TAG}_pmc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_mmr_f16.py 5 > $O/${TAG}_kt.log 2>&1
f=$(find $O/${TAG}_kt -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-160
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc -- python3 $R/scratch/time_mmr_f16.py 2 > $O/${TAG}_pmc.log 2>&1
python3 $R/scratch/pmc_dump.py gram_mfma $O/${TAG}_pmc 2>/dev/null | tail -12
