#!/bin/bash
# after `gpurun -- bash scratch/collect_r04.sh r04`: condense gpurun_out/r04_* into profiles/r04_*
cd "$(dirname "$0")/.." && G=gpurun_out
python3 profiles/summarize.py r04 $G/r04_head_kt $G/r04_head_fetch $G/r04_head_write "python3 bench.py --steps 30 --warmup 3 --no-cpu --no-extras" > /dev/null
python3 profiles/summarize.py kernels r04_shard $G/r04_shard_kt "python3 scratch/time_shard_step.py  (one GPU's share of the headline at 8 GPUs: 1.25 M x 768 f32 through ShardedIndex.search_topk at world 1)" "scan_fixed,tail_stage1,tail_stage2,merge_topk,stage_query" > /dev/null
cp $G/r04_shard_timeline.txt profiles/r04_shard_timeline.txt
python3 profiles/summarize.py kernels r04_batch256_image $G/r04_b256_kt "python3 bench.py --batch 256 --image --steps 12 --warmup 2 --no-cpu --settle-ms 0" "gemm8_kernel,batch_select,batch_band,batch_rescore,batch_emit,prep_queries" --pmc $G/r04_b256_fetch $G/r04_b256_write > /dev/null
python3 profiles/summarize.py kernels r04_batch256 $G/r04_b256p_kt "python3 bench.py --batch 256 --steps 8 --warmup 2 --no-cpu --settle-ms 0  (the default batched path of an f32 index: no nomination image)" "gemm_nominate_kernel,batch_select,batch_band,batch_rescore,batch_emit,prep_queries" --pmc $G/r04_b256p_fetch $G/r04_b256p_write > /dev/null
python3 profiles/summarize.py kernels r04_c2 $G/r04_c2_kt "python3 scratch/time_c2_abi.py  (rlr_engine_search_with_diversity: 100 k x 768 f32, top_k 100, lambda 0.3)" "mmr_greedy,scan_fixed,gram_tiled,tail_stage1,tail_stage2,stage_query,pool_prepare" > /dev/null
python3 profiles/summarize.py kernels r04_c2_text $G/r04_hyb_kt "python3 scratch/time_c2_hybrid.py hybrid-only  (search / search_with_diversity with the query text: 100 k x 768 f32, top_k 100, lambda 0.3, GPU BM25)" "mmr_greedy,scan_fixed,gram_tiled,hybrid_pool,hybrid_emit,lex_unpack,score_rows_staged,tail_stage1,tail_stage2,bm25_terms,lex_sample,lex_filter,lex_final,lex_clear,stage_query" > /dev/null
python3 profiles/summarize.py kernels r04_c5_share $G/r04_c5_kt "python3 scratch/time_c5_shard.py --image  (6.25 M x 1024 binary16, 1024 queries, pool 308, MMR 0.7; steady state: second full-size pass)" "gemm8_kernel,gram_mfma_f32,batch_rescore,mmr_greedy,batch_band,batch_emit,batch_select" --pmc $G/r04_c5_fetch $G/r04_c5_write $G/r04_c5_sq $G/r04_c5_sq2 $G/r04_c5_sq3 > /dev/null
for f in bench_under_rocprof bench_batch256_image_under_rocprof bench_batch256_under_rocprof bench_n1 bench_inprocess_n1; do cp $G/r04_$f.json profiles/r04_$f.json; done
ls -la profiles/r04_*
# (scratch/collect_r04_traffic.sh: the scan's counter traffic at the two per-GPU share shapes)
if [ -d $G/r04_shard_fetch ] && [ -d $G/r04_c4_fetch ]; then
python3 profiles/summarize.py r04_shard1of8 $G/r04_shard_kt $G/r04_shard_fetch $G/r04_shard_write "python3 scratch/time_shard_step.py  (1.25 M x 768 f32: one GPU's share of the headline at 8 GPUs)" > /dev/null
python3 profiles/summarize.py r04_c4share $G/r04_c4_kt $G/r04_c4_fetch $G/r04_c4_write "python3 bench.py --rows 12500000 --steps 20 --warmup 3 --no-cpu --no-extras --settle-ms 500  (12.5 M x 768 f32: one GPU's share of config 4)" > /dev/null
fi
