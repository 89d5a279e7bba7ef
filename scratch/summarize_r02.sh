#!/bin/bash
# after `gpurun -- bash scratch/collect_r02.sh r02`: condense gpurun_out/r02_* into profiles/r02_*
cd "$(dirname "$0")/.." && G=gpurun_out
python3 profiles/summarize.py r02 $G/r02_head_kt $G/r02_head_fetch $G/r02_head_write "python3 bench.py --steps 30 --warmup 3 --no-cpu --no-extras" > /dev/null
python3 profiles/summarize.py kernels r02_batch256_image $G/r02_b256_kt "python3 bench.py --batch 256 --image --steps 12 --warmup 2 --no-cpu --settle-ms 0" "gemm8_kernel,batch_band,batch_rescore,batch_emit,batch_tighten,collect_packed,hist1,hist2,prep_queries" --pmc $G/r02_b256_sq1 $G/r02_b256_sq2 $G/r02_b256_fetch > /dev/null
python3 profiles/summarize.py kernels r02_c2 $G/r02_c2_kt "python3 scratch/time_c2_abi.py  (rlr_engine_search_with_diversity: 100 k x 768 f32, top_k 100, lambda 0.3)" "mmr_greedy,scan_fixed,gram_tiled,rescore_staged,pool_prepare,sort_emit,collect_find2,hist2_find1,diverse_emit" > /dev/null
python3 profiles/summarize.py kernels r02_c2_text $G/r02_hyb_kt "python3 scratch/time_c2_hybrid.py hybrid-only  (search / search_with_diversity with the query text: 100 k x 768 f32, top_k 100, lambda 0.3, GPU BM25)" "mmr_greedy,scan_fixed,gram_tiled,hybrid_pool,hybrid_emit,lex_unpack,score_rows_staged,sort_emit,collect_find2,rescore_staged,hist2_find1,bm25_term,lex_sample,lex_filter,lex_final,lex_select_pass,lex_collect,lex_clear,lex_sort" > /dev/null
python3 profiles/summarize.py kernels r02_c5_share $G/r02_c5_kt "python3 scratch/time_c5_shard.py --image  (6.25 M x 1024 binary16, 1024 queries, pool 308, MMR 0.7; steady state: second full-size pass)" "gemm8_kernel,gram_tiled,batch_rescore,collect_packed,hist1,mmr_greedy,batch_tighten,hist2,batch_band,batch_emit" > /dev/null
python3 profiles/summarize.py kernels r02_lexical $G/r02_lex_kt "python3 scratch/time_lexical.py 200000  (GPU BM25, 200 k chunks x 30 tokens)" "bm25_term,lex_sample,lex_filter,lex_final,lex_select_pass,lex_collect,lex_sort,lex_clear" > /dev/null
python3 profiles/summarize.py kernels r02_batch8 $G/r02_multi8_kt "python3 bench.py --batch 8 --steps 20 --warmup 3 --no-cpu --settle-ms 0  (8 queries share one scan)" "scan_multi,collect_packed,hist1,hist2,batch_rescore,batch_band,batch_emit" > /dev/null
for f in bench_under_rocprof bench_batch256_image_under_rocprof bench_batch8_under_rocprof bench_n1; do cp $G/r02_$f.json profiles/r02_$f.json; done
ls -la profiles/r02_*
