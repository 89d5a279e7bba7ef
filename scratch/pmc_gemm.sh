#!/bin/bash
# SQ counter passes over the batched image GEMM (one rocprofv3 run per pass, counters only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --batch 256 --image --steps 4 --warmup 1 --no-cpu --settle-ms 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  (cd $R && timeout -k 10 280 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_gemm_$i -- $CMD > $R/gpurun_out/pmc_gemm_$i.log 2>&1) || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_gemm_$i.log; exit 1; }
  echo "pass $i done"
done
python3 $R/scratch/pmc_dump.py gemm_image_kernelILb0 $R/gpurun_out/pmc_gemm_1 $R/gpurun_out/pmc_gemm_2 $R/gpurun_out/pmc_gemm_3 | tee $R/gpurun_out/pmc_gemm_summary.txt
