#!/usr/bin/env python3
"""Benchmark of the search_documents hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one query: 768-d cosine top-100 over a 10M-chunk f32 corpus (the
configuration BASELINE.json's metric is quoted on), end to end through the C ABI:
scan -> select -> reference-order re-score -> sort -> results on the host.  The corpus
is generated on the device (rlr_index_fill_synthetic), so the timed region starts with
every input resident in HBM.  With N > 1 (launched by torch.distributed.run, one process
per GPU) the same 10M rows are sharded N ways and every step adds the RCCL all-gather of
the per-shard partial top-k and the merge -- total work is fixed => "scaling": "strong".

Rank 0 prints ONE JSON line: metric/value/unit/... + "roofline" (with the measured read / copy peaks of the box beside
the spec peak) + "cpu_baseline" + (N = 1 only, measured outside the headline's timed region) "configs": one GPU's share of
the headline at 8 GPUs through the sharded code path (headline_shard_1of8, with implied_speedup_8), BASELINE configs C2 /
C3 (with and without the nomination image) / C4-share / C5-share, and the headline on hostile score distributions
(tight clusters, exact duplicates), each with its own roofline object.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense binary16 / bf16 MFMA peak (same guide; never the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows (total over all GPUs)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--seed", type=int, default=0x5EED0003)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the same corpus the CPU baseline scans")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--check", type=int, default=1, help="queries verified against the oracle on the CPU sample")
    ap.add_argument("--no-profile", action="store_true",
                    help="do not record HIP events in the timed region (roofline fields become null)")
    ap.add_argument("--image", action="store_true",
                    help="keep the binary16 nomination image for the batched path (rlr_index_enable_batch_image)")
    ap.add_argument("--settle-ms", type=float, default=2500.0,
                    help="untimed clock/power settle phase before the warmup steps (0 disables).  The scan runs 1.2-1.6 %% "
                         "slower during the first second of sustained load than from the second second on "
                         "(scratch/warm_drift.py: 0.8645 of the HBM peak, then 0.8785 +- 0.0003 for 19 s)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational measurements outside the timed region (optional_modes, configs)")
    ap.add_argument("--image-scan", action="store_true",
                    help="single queries nominate over the binary16 image too (half the scan bytes; same results)")
    ap.add_argument("--q8-scan", action="store_true",
                    help="single queries nominate over the 8-bit copy (a quarter of the scan bytes; same results)")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process driving --gpus N devices through rlr_multi (persistent shard workers, ncclAllGather of the "
                         "partial top-k lists + merge kernel behind the C ABI) instead of one rank per GPU over torch.distributed: "
                         "the form a single-process Rust server uses (INTEGRATION.md section 5)")
    ap.add_argument("--batch", type=int, default=1,
                    help="queries per step; >= 16 takes the matrix-core (MFMA) batched path (BASELINE config 3 uses 256)")
    return ap.parse_args()


def queries_without_oracle(rlr, dim, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(n)])


def q8_kernel_name(dim):
    """which 8-bit scan kernel csrc/q8.hip launches for this row width (launch_q8_scan)"""
    packed = os.environ.get("RLR_Q8_PACKED", "1")[:1] != "0" and dim in (128, 256, 384, 512, 768, 1536)
    return "q8_scan_packed_kernel" if packed else "q8_scan_kernel"


def batch_kernel_name(dim, nq, image):
    """which GEMM csrc/gemm.hip launches for a batch (launch_gemm_nominate's dispatch)"""
    if not image:
        return "gemm_nominate_kernel"
    if nq <= int(os.environ.get("RLR_GEMM_RESIDENT_MAX", "128")) and dim % 256 == 0 and dim <= 1152:
        return "gemm_resident_kernel"
    return "gemm8_kernel" if dim % 128 == 0 else "gemm_nominate_kernel"


def pmc_traffic(bytes_per_launch, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of
    this same command (profiles/rNN_pmc.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    separate passes).  PMC counters cannot be read from inside the process, so the figure is
    quoted from the newest summary of the SAME kernel whose byte count matches this run's shape
    (within 5 %); any other shape -- a shard of an N > 1 run, other dims -- reports null.
    -> (bytes, file, collected on this very build?)"""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            d = json.load(open(f))
            t = float(d["pmc"]["hbm_bytes_per_launch"])
            name = str(d.get("kernel", ""))
        except Exception:
            continue
        if kernel in name and abs(t - bytes_per_launch) <= 0.05 * bytes_per_launch:
            best = (t, os.path.basename(f), d.get("build_source_sha16") == source_sha16())
    return best


def batched_traffic(bytes_per_launch, kernel, tag):
    """HBM bytes of the batch's main GEMM pass from the committed rocprofv3 counter passes of the same workload
    (profiles/rNN_<tag>_kernels.json, `counters_by_ordinal`: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE per launch of the
    LAST ordinal inside a batch = the filtered main pass); the newest round's summary.  PMC counters cannot be read from
    inside the process; a figure outside [0.9, 4] x the operand bytes would mean another shape and is not quoted.
    -> (bytes, description, collected on this very build?)"""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_kernels.json"))):
        try:
            doc = json.load(open(f))
            cbo = doc.get("counters_by_ordinal", {})
        except Exception:
            continue
        for name, by_ord in cbo.items():
            if kernel not in name or "<true" in name or not by_ord:
                continue
            o = max(by_ord, key=int)
            c = by_ord[o]
            fetch = c.get("FETCH_SIZE", {}).get("bytes_corrected_x2")
            if fetch is None or not (0.9 * bytes_per_launch <= fetch <= 4.0 * bytes_per_launch):
                continue
            write = c.get("WRITE_SIZE", {}).get("bytes", 0.0)
            best = (fetch + write, os.path.basename(f) + f" ({name}, launch {o} of a batch)",
                    doc.get("build_source_sha16") == source_sha16())
    return best


def quote_traffic(roof, t, what):
    """Counter traffic into a roofline object: `traffic` only when the committed summary was collected on THIS build of the
    library (csrc/ + include/ hash equal); a summary of another build goes under `profiled_traffic` with a note, and
    `traffic` stays null -- a fresh kernel time is never paired with an older kernel's bytes."""
    roof["traffic"], roof["traffic_source"] = None, None
    if not t:
        return
    src = f"profiles/{t[1]}: {what}"
    if t[2]:
        roof["traffic"], roof["traffic_source"] = t[0], src
    else:
        roof["profiled_traffic"] = {"value": t[0], "source": src,
                                    "stale_profile": "collected on another build of csrc/ + include/ than the one running"}


def batched_roofline(prof, dim, nq, image, tag="batch256_image"):
    """roofline object of the dominant launch of a batch (the filtered main GEMM pass), from the library's HIP
    events on its own stream.  `bound` is the roof with the larger ideal time for this launch: the operand bytes
    once at the HBM peak against 2*Q*rows*dim flops at the dense binary16 MFMA peak (both reported)."""
    n = max(prof.n_batches, 1)
    ms, b, fl = prof.batch_main_ms / n, prof.batch_main_bytes / n, prof.batch_main_flops / n
    if ms <= 0:
        return None
    gbps, tflops = b / (ms * 1e-3) / 1e9, fl / (ms * 1e-3) / 1e12
    t_hbm, t_mfma = b / (HBM_PEAK_GBPS * 1e9), fl / (MFMA_F16_PEAK_TFLOPS * 1e12)
    hbm = {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS}
    mfma = {"achieved": tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_F16_PEAK_TFLOPS}
    main = dict(mfma if t_mfma >= t_hbm else hbm)
    traffic = batched_traffic(b, batch_kernel_name(dim, nq, image), tag)
    main["bound"] = "mfma" if t_mfma >= t_hbm else "hbm"
    quote_traffic(main, traffic, "FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this shape")
    main.update({"kernel": batch_kernel_name(dim, nq, image) + " (the filtered main pass of a batch)",
                 "kernel_ms": ms, "bytes_per_launch": b, "flops_per_launch": fl,
                 "hbm": hbm, "mfma": mfma,
                 "all_gemm_launches_ms": prof.batch_gemm_ms / n,
                 "note": "binary16 MFMA nominates; emitted rows and scores are re-scored in f32 reference order"})
    return main


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, rlr):
    """Times the oracle (CPU port of the reference loops) on a bounded sample of the same
    corpus: rows [0, cpu_rows) of the synthetic stream, full search (scan + stable sort +
    take), 1 thread -- the reference's actual behaviour -- the scan alone (=> the sort's share), the same
    arithmetic with rows split over all host cores, and the literal O(k^2 P) mmr_diversify loop on one
    pool of 300 (rag_engine.rs:788-835).  Search figures are extrapolated linearly to the full row count."""
    from oracle import oracle as O  # checker / reported baseline only

    n = min(args.cpu_rows, args.rows)
    t0 = time.perf_counter()
    rows = O.synth_rows(n, args.dim, args.seed, f16=(args.dtype == "f16"))
    gen_s = time.perf_counter() - t0
    q = O.synth_query(args.dim, args.seed + 1)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        r1 = O.search(rows, q, args.k)
    t1 = (time.perf_counter() - t0) / reps
    cores = os.cpu_count() or 1
    qn = O.normalize(q)
    t0 = time.perf_counter()
    for _ in range(reps):
        O.scan(rows, qn, threads=1)
    ts = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        O.scan(rows, qn, threads=cores)
    tm = (time.perf_counter() - t0) / reps
    # literal MMR loop: pool = the 300 best of the sample, k = 100, lambda = 0.3 (config C2's shape)
    pool = O.search(rows, q, 300)
    emb = np.ascontiguousarray(rows[np.asarray(pool[0], dtype=np.int64)])
    t0 = time.perf_counter()
    O.mmr(emb, np.asarray(pool[1], dtype=np.float32), 100, 0.3)
    t_mmr = time.perf_counter() - t0
    scale = args.rows / n
    return {
        "value": 1.0 / (t1 * scale),
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "cpu_model": cpu_model(),
        "sample": (f"oracle search (scan + stable sort + take) of 1 query over rows [0,{n}) of the same synthetic "
                   f"corpus, {t1 * 1e3:.0f} ms/query on 1 thread, extrapolated x{scale:.0f} to {args.rows} rows; "
                   f"contiguous matrix, no per-candidate clone (both favour the reference)"),
        "scan_only_ms_on_sample": ts * 1e3,
        "sort_and_take_ms_on_sample": max(t1 - ts, 0.0) * 1e3,
        "sort_note": "full stable sort of all scored chunks as rag_engine.rs:543 does; = search - scan on the sample",
        "mmr_1_thread": {"ms": t_mmr * 1e3, "pool": 300, "top_k": 100, "lambda": 0.3,
                         "note": "literal mmr_diversify loop (rag_engine.rs:788-835), 1 156 650 dot products"},
        "all_cores": {"value": 1.0 / (tm * scale), "cores": cores,
                      "note": "scan only, rows split over threads; not something the reference does"},
        "sample_gen_s": round(gen_s, 2),
    }, rows, r1


def source_sha16():
    """sha256 over the sources librlr_gpu.so is built from (csrc/ + include/), first 16 hex digits"""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "rust-local-rag_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p):
                h.update(f.encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def ensure_built():
    """Bring librlr_gpu.so up to date with csrc/ and include/ (build.py recompiles only what is stale: a no-op
    when the shipped binary is fresh; hipcc is a child process, nothing here touches the GPU) -- one rank
    builds, the others wait on the lock."""
    import fcntl

    import __graft_entry__

    with open(os.path.join(ROOT, "rust-local-rag_amd", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        __graft_entry__._load_build_module().build()


# ------------------------------------------------------------------------------------------------------
# BASELINE configs measured beside the headline (N = 1, outside its timed region)
# ------------------------------------------------------------------------------------------------------
def config_c3(rlr, ix, args, torch):
    """C3: 10 M x 768 f32, 256 batched queries, top-100 (same index as the headline), both ways a caller can run it:
    `without_image` -- rlr_search_topk on the f32 index as it is (gemm_nominate_kernel streams the f32 rows and rounds
    them to binary16 on the way into the matrix cores) -- and over the opt-in nomination image (+dim*2 B/row of HBM,
    gemm8_kernel streams half the bytes).  Identical results; the top-level value is the image's."""
    nq, steps = 256, 12
    pool = queries_without_oracle(rlr, args.dim, nq + steps + 3, args.seed + 3)

    def leg(image):
        ix.enable_batch_image(image)
        try:
            for i in range(3):
                ix.search_topk(pool[i:i + nq], args.k)
            ix.profile_read(reset=True)
            ix.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                last = ix.search_topk(pool[3 + i:3 + i + nq], args.k)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ix.profile_enable(False)
            p = ix.profile_read()
        finally:
            ix.enable_batch_image(False)
        nb = max(p.n_batches, 1)
        return {"value": steps * nq / el, "unit": "queries/s", "ms_per_batch": el / steps * 1e3,
                "stages_ms": {"gemm_all_launches": p.batch_gemm_ms / nb, "select_and_finish": p.batch_other_ms / nb},
                "fallback_queries": p.n_batch_fallbacks,
                "n_batches_without_image": p.n_batches_without_image,   # rlr_profile's hint that an image would have paid
                "extra_hbm_bytes": len(ix) * args.dim * 2 if image else 0,
                "roofline": batched_roofline(p, args.dim, nq, image, "batch256_image" if image else "batch256")}, last

    plain, r0 = leg(False)
    img, r1 = leg(True)
    same = bool(np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1].view(np.uint32), r1[1].view(np.uint32)))
    out = {"workload": f"C3: {len(ix)} chunks x {args.dim}-d f32, {nq} batched queries/step, top_k={args.k}, "
                       f"nomination image (+dim*2 B/row)"}
    out.update(img)
    out["without_image"] = plain
    out["without_image"]["what"] = ("the default path of an f32 index: rlr_search_topk with 256 queries, no opt-in "
                                    "(gemm_nominate_kernel over the f32 rows)")
    out["image_speedup"] = img["value"] / plain["value"] if plain["value"] else None
    out["same_results_both_ways_on_the_last_batch"] = same
    return out


def c2_hybrid_leg(rlr, torch, ix, qs, n, dim, k, lam, steps):
    """C2 the way the reference's search_documents runs it: with the query TEXT, whose BM25 scores (`LexicalIndex::score`,
    5 x pool of them, rag_engine.rs:505) are blended into the pool before MMR.  rlr_engine_search_text at the C ABI: BM25
    on its own stream beside the scan, blend + cut + MMR in the same enqueue.  Synthetic chunk texts: 40 Zipf-distributed
    words of a 20 000-word vocabulary per chunk, 6-word queries."""
    import ctypes as C

    N = rlr._native
    lex_mod = importlib.import_module("rust-local-rag_amd.lexical")
    rng = np.random.default_rng(0x5EED0012)
    V = 20000
    vocab = np.array([f"t{i:05d}" for i in range(V)])
    zipf = 1.0 / np.arange(1, V + 1)
    zipf /= zipf.sum()
    lx = lex_mod.LexicalIndex(0)
    try:
        for b0 in range(0, n, 10000):
            words = rng.choice(V, size=(min(10000, n - b0), 40), p=zipf)
            for i in range(words.shape[0]):
                lx.add_tokens(b0 + i, vocab[words[i]])
        toks = [" ".join(vocab[rng.choice(V, size=6, p=zipf)]).encode() for _ in range(steps + 20)]
        cap = max(3 * k, k + 10)
        hits = (N.SearchHitC * cap)()
        nn = C.c_uint32()
        L = N.lib()

        def call(i):
            q = qs[i]
            N.check(L.rlr_engine_search_text(ix.handle, lx._h, q.ctypes.data_as(N.f32p), dim, toks[i], len(toks[i]), k, lam, 0,
                                             None, hits, cap, C.byref(nn)))

        for i in range(20):
            call(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            call(20 + i)
        el = time.perf_counter() - t0
        n_lex = sum(1 for j in range(nn.value) if hits[j].lexical_score != 0.0)
    finally:
        lx.close()
    return {"what": "rlr_engine_search_text: the same search with the query text (GPU BM25 beside the scan, hybrid blend, "
                    "MMR; one synchronisation)", "value": steps / el, "unit": "queries/s", "ms_per_query": el / steps * 1e3,
            "results_of_the_last_query": int(nn.value), "of_them_with_a_lexical_score": n_lex}


def config_c2(rlr, torch):
    """C2: 100 k x 768 f32, single query, top-100, MMR lambda 0.3: rlr_engine_search_with_diversity timed at the C ABI
    (what a Rust host calls; the Python veneer's per-result objects are not part of the path)."""
    import ctypes as C

    N = rlr._native
    n, dim, k, lam, steps = 100_000, 768, 100, 0.3, 200
    ix = rlr.GpuIndex(dim)
    try:
        ix.fill_synthetic(n, seed=0x5EED0002, n_clusters=200)
        rng = np.random.default_rng(0x5EED0002)
        qs = rng.standard_normal((steps + 20, dim)).astype(np.float32)      # raw embeddings: the engine normalises (:494)
        cap = max(3 * k, k + 10)
        hits = (N.SearchHitC * cap)()
        nn = C.c_uint32()
        L = N.lib()

        def call(q):
            N.check(L.rlr_engine_search_with_diversity(ix.handle, q.ctypes.data_as(N.f32p), dim, k, lam, None, None, None, 0,
                                                       hits, cap, C.byref(nn)))

        for i in range(20):
            call(qs[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            call(qs[20 + i])
        el = time.perf_counter() - t0
        ix.profile_read(reset=True)
        ix.profile_enable(True)                                              # a second pass for the per-kernel times
        for i in range(50):
            call(qs[20 + i])
        ix.profile_enable(False)
        p = ix.profile_read()
        hybrid = c2_hybrid_leg(rlr, torch, ix, qs, n, dim, k, lam, steps)
    finally:
        ix.close()
    ns = max(p.n_scan_launches, 1)
    scan_ms = p.scan_ms / ns
    kern = {"scan": scan_ms, "select": p.select_ms / ns, "rescore_sort": p.rescore_ms / ns,
            "pool_gather_gram_greedy_emit": p.mmr_ms / max(p.n_mmr, 1)}
    b = n * dim * 4
    gbps = b / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return {"workload": f"C2: {n} chunks x {dim}-d f32, 1 query/step, top_k={k}, MMR lambda={lam} (pool 300 -> {k}); "
                        f"rlr_engine_search_with_diversity at the C ABI (search -> MMR fused on the device)",
            "value": steps / el, "unit": "queries/s", "ms_per_query": el / steps * 1e3,
            "kernels_ms": kern, "kernel_sum_ms": sum(kern.values()), "results_per_query": int(nn.value),
            "with_query_text": hybrid,
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": gbps / HBM_PEAK_GBPS, "traffic": None, "kernel": "scan_fixed_kernel",
                         "kernel_ms": scan_ms, "bytes_per_launch": b,
                         "note": "307 MB per launch: launch-latency-limited, not bandwidth-limited (SURVEY 8(d))"}}


def config_c5_share(rlr, torch):
    """One GPU's share of C5: 6.25 M x 1024 binary16 rows (50 M / 8), 1024 batched queries, top-100, MMR 0.7."""
    n, dim, nq, k, lam = 6_250_000, 1024, 1024, 100, 0.7
    eng = rlr.RagEngine(dim, "f16")
    try:
        eng.index.fill_synthetic(n, seed=0x5EED0005, n_clusters=500)
        eng._chunks = [None] * n
        eng.index.enable_batch_image(True)
        qs = queries_without_oracle(rlr, dim, nq, 0x5EED0005)
        ix = eng.index
        pool = max(3 * k, k + 10)
        # one untimed full-size pass first: it grows the per-call workspaces (hipMalloc / hipHostMalloc of the candidate
        # lists, the 1024 x 300 x 300 Gram block, the pinned staging) -- a server pays that once, not per batch
        r, c = ix.search_topk(qs, pool + 8)
        sc = (np.float32(0.7) * c[:, :pool]).astype(np.float32)
        rows_in, sizes = np.ascontiguousarray(r[:, :pool]), np.full(nq, pool, np.uint32)
        ix.mmr_select_batch(rows_in, sc, sizes, k, lam)
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        reps = 3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r, c = ix.search_topk(qs, pool + 8)                   # what search_with_diversity_batch fetches
        torch.cuda.synchronize()
        t_search = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.mmr_select_batch(rows_in, sc, sizes, k, lam)
        torch.cuda.synchronize()
        t_mmr = (time.perf_counter() - t0) / reps
        ix.profile_enable(False)
        p = ix.profile_read()
    finally:
        eng.close()
    return {"workload": f"C5 per-GPU share: {n} chunks x {dim}-d binary16, {nq} batched queries, top_k={k}, "
                        f"MMR lambda={lam} (pool {pool}); nomination image",
            "value": nq / (t_search + t_mmr), "unit": "queries/s",
            "stages_ms": {"batched_search_pool308": t_search * 1e3, "batched_mmr": t_mmr * 1e3,
                          "gemm_all_launches": p.batch_gemm_ms / reps, "select_and_finish": p.batch_other_ms / reps,
                          "mmr_kernels": p.mmr_ms / reps},
            "timed_passes": reps, "fallback_queries": p.n_batch_fallbacks, "roofline": batched_roofline(p, dim, nq, True, "c5_share")}


def config_shard_1of8(rlr, torch, sharded, args, headline_ms):
    """One GPU's share of the headline at 8 GPUs -- 1.25 M x 768 f32, single query, top-100 -- through the SHARDED code
    path at world 1: rlr_search_topk_device_begin (pipelines enqueued on torch's stream) -> [the all-gather sits here at
    world > 1] -> merge kernel (results into pinned host memory) -> _end.  What it leaves out of an 8-GPU step is the
    8-rank RCCL all-gather of 8 x 100 x 8 B (latency-bound, tens of us).  implied_speedup_8 = headline ms / this ms."""
    n, steps = args.rows // 8, 400
    qs = queries_without_oracle(rlr, args.dim, steps + 40, args.seed + 8)
    sh = sharded.ShardedIndex(args.dim, n, args.dtype, device=0, rank=0, world=1)
    try:
        sh.fill_synthetic(args.seed)
        ix = sh.index
        t_settle = time.perf_counter()           # the same untimed settle phase as the headline's (see --settle-ms)
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            sh.search_topk(qs[0], args.k)
        for i in range(40):
            sh.search_topk(qs[i], args.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sh.search_topk(qs[40 + i], args.k)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for i in range(steps):
            ix.search_topk(qs[40 + i], args.k)                      # the unsharded call on the same shard, for reference
        plain = (time.perf_counter() - t0) / steps
        ix.profile_read(reset=True)
        ix.profile_enable(True)                                      # second pass: per-stage HIP events
        sh.time_exchange(True)
        for i in range(100):
            sh.search_topk(qs[40 + i], args.k)
        ix.profile_enable(False)
        p = ix.profile_read()
        merge_ms = sh.exchange_ms_per_step()
        sh.time_exchange(False)
    finally:
        sh.index.close()
    ns = max(p.n_scan_launches, 1)
    scan_ms = p.scan_ms / ns
    elem = 2 if args.dtype == "f16" else 4
    b = n * args.dim * elem
    gbps = b / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return {"workload": f"headline shard 1 of 8: {n} chunks x {args.dim}-d {args.dtype}, 1 query/step, top_k={args.k}; "
                        f"ShardedIndex.search_topk at world 1 (begin -> merge kernel -> end)",
            "value": 1.0 / el, "unit": "queries/s", "ms_per_step": el * 1e3,
            "stages_ms": {"scan": scan_ms, "select": p.select_ms / ns, "rescore_sort": p.rescore_ms / ns, "merge": merge_ms},
            "stage_note": "HIP events of a second pass: select = tail stage 1 (bin search + collect + re-score, or the digit-2 "
                          "histogram), rescore_sort = tail stage 2 (sort + emit), merge = all-gather slot + merge kernel",
            "fixed_cost_us": (el * 1e3 - scan_ms) * 1e3,
            "plain_search_topk_ms": plain * 1e3,
            "implied_speedup_8": headline_ms / (el * 1e3),
            "implied_note": "headline ms_per_step / this; an 8-rank all-gather of 6.4 KB is not in it",
            "candidates_per_query": p.n_candidates / max(p.n_searches, 1), "band_retries": p.n_retries,
            "roofline": scan_roofline(gbps, scan_ms, b)}


def scan_roofline(gbps, scan_ms, b):
    """roofline object of a single-query configuration: the scan kernel against the HBM peak; `traffic` from the committed
    counter passes of this kernel at this shape (profiles/r*_pmc.json), under quote_traffic's same-build rule"""
    roof = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "traffic": None, "kernel": "scan_fixed_kernel", "kernel_ms": scan_ms, "bytes_per_launch": b}
    quote_traffic(roof, pmc_traffic(b, "scan_fixed_kernel"),
                  "rocprofv3 --pmc passes of this kernel over an index of this shape, FETCH_SIZE x2 + WRITE_SIZE")
    return roof


def config_single(rlr, torch, args, n, seed, n_clusters, what, steps=30):
    """single-query top-k over a fresh index of n rows (C4's per-GPU share, the hostile score distributions)"""
    qs = queries_without_oracle(rlr, args.dim, steps + 10, seed + 1)
    ix = rlr.GpuIndex(args.dim, args.dtype)
    try:
        ix.fill_synthetic(n, seed=seed, n_clusters=n_clusters)
        t_settle = time.perf_counter()           # the same untimed settle phase as the headline's (see --settle-ms)
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            ix.search_topk(qs[0], args.k)
        for i in range(10):
            ix.search_topk(qs[i], args.k)
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ix.search_topk(qs[10 + i], args.k)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        ix.profile_enable(False)
        p = ix.profile_read()
    finally:
        ix.close()
    ns = max(p.n_scan_launches, 1)
    scan_ms = p.scan_ms / ns
    b = n * args.dim * (2 if args.dtype == "f16" else 4)
    gbps = b / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return {"workload": what, "value": 1.0 / el, "unit": "queries/s", "ms_per_step": el * 1e3,
            "stages_ms": {"scan": scan_ms, "select": p.select_ms / ns, "rescore_sort": p.rescore_ms / ns},
            "candidates_per_query": p.n_candidates / max(p.n_searches, 1), "band_retries": p.n_retries,
            "roofline": scan_roofline(gbps, scan_ms, b)}


def in_process(args):
    """headline workload, rows sharded over args.gpus devices of THIS process (rlr_multi, RCCL exchange)"""
    import gc

    rlr = importlib.import_module("rust-local-rag_amd")
    if rlr.device_count() < args.gpus:
        raise SystemExit(f"--in-process --gpus {args.gpus}: only {rlr.device_count()} device(s) visible")
    qs = queries_without_oracle(rlr, args.dim, args.warmup + args.steps, args.seed)
    mi = rlr.MultiGpuIndex(args.dim, list(range(args.gpus)), args.dtype)
    t0 = time.perf_counter()
    mi.fill_synthetic(args.rows, args.seed)
    fill_s = time.perf_counter() - t0
    mi.set_exchange("rccl")
    gc.collect()
    gc.freeze()
    t_settle = time.perf_counter()
    while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        mi.search_topk(qs[0], args.k)
    for i in range(args.warmup):
        mi.search_topk(qs[i], args.k)
    mi.stats(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = mi.search_topk(qs[args.warmup + i], args.k)   # returns after the merged result is in host memory
    elapsed = time.perf_counter() - t0
    st = mi.stats()
    elem = 2 if args.dtype == "f16" else 4
    print(json.dumps({
        "metric": "queries/sec + achieved HBM GB/s, 768-d cosine top-100 over 10M chunks",
        "value": args.steps / elapsed, "unit": "queries/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.rows} chunks x {args.dim}-d {args.dtype}, 1 query per step, top_k={args.k}",
                   "parallelism": f"one process, rlr_multi over {args.gpus} device(s), ncclAllGather + merge kernel behind the C ABI"},
        "aggregate_hbm_GBps": args.rows * args.dim * elem / (elapsed / args.steps) / 1e9,
        "exchange": {"rccl_calls": st["n_topk_rccl"], "host_merge_calls": st["n_topk_host_merge"],
                     "rccl_fell_back": st["n_topk_rccl_fell_back"],
                     "rccl_call_ms": st["topk_rccl_ms"] / max(st["n_topk_rccl"], 1)},
        "fill_s": fill_s, "top1_row": int(last[0][0][0]) if last[0].size else None}))
    mi.close()


def main():
    args = parse()
    ensure_built()
    if args.in_process:
        return in_process(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet);
        # one rank per GPU over RCCL, as the driver's own `torch.distributed.run` command line does
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.call(cmd))

    import torch

    rlr = importlib.import_module("rust-local-rag_amd")
    if rlr.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: librlr_gpu.so has no CPU path")
    if os.environ.get("RLR_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # rehearsal of the N > 1 code path on a one-GPU box (with RLR_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("RLR_BENCH_FORCE_DIST") == "1"  # rehearse RCCL init + all-gather with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("RLR_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    sharded = importlib.import_module("rust-local-rag_amd.sharded")

    # The oracle is only the checker / reported baseline and only rank 0 of a single-GPU run uses it;
    # the timed queries come from a seeded numpy generator so every rank sees the same ones.
    O = None
    if world == 1 and not args.no_cpu:
        try:
            from oracle import oracle as O
            O.lib()
        except Exception:
            O = None
    n_q = args.warmup + args.steps
    if args.batch > 1:
        pool = queries_without_oracle(rlr, args.dim, args.batch + n_q, args.seed)
        qs = [pool[i:i + args.batch] for i in range(n_q)]  # a sliding window: every step a different batch
    else:
        qs = queries_without_oracle(rlr, args.dim, n_q, args.seed)

    # ---- corpus: generated in HBM, sharded by contiguous row ranges -------------------
    t0 = time.perf_counter()
    sh = sharded.ShardedIndex(args.dim, args.rows, args.dtype, device=local_rank, rank=rank, world=world)
    sh.fill_synthetic(args.seed)
    fill_s = time.perf_counter() - t0
    ix = sh.index
    n_local = len(ix)
    if args.image or args.image_scan or args.q8_scan:
        ix.enable_batch_image(args.image or args.image_scan, single_query=args.image_scan, q8=args.q8_scan)

    # Measured-peak denominators ("re-measure on the box"), once, before anything is timed: a read-only
    # stream and a device-to-device copy over this rank's own rows (rlr_index_probe_bandwidth).
    measured = None
    if rank == 0 and not args.no_profile:
        try:
            rd, rd_ms = ix.probe_bandwidth(0, 5)
            cp, cp_ms = ix.probe_bandwidth(1, 5)
            measured = {"read": rd, "read_ms": rd_ms, "copy": cp, "copy_ms": cp_ms}
        except Exception as e:  # the probe must never take the headline down
            measured = {"error": str(e)}

    force_sharded = os.environ.get("RLR_BENCH_FORCE_SHARDED") == "1" or force_dist  # rehearse the N>1 code path on one GPU
    use_sharded = world > 1 or force_sharded

    def step(i):
        if not use_sharded:
            return ix.search_topk(qs[i], args.k)
        return sh.search_topk(qs[i], args.k)

    # Untimed settle phase before the W warmup steps: a cold GPU needs some hundred ms of work before clocks and the
    # HBM power state are at their sustained level.  (Round 1 also ran >= 256 calls here to keep "a one-off 30-40 ms
    # HIP-runtime stall around call 206" out of the timed region.  A HIP API trace of 400 steps shows that stall
    # BETWEEN two API calls, not inside one, and gc.callbacks puts the start of CPython's first full (generation-2)
    # collection -- over the object graph `import torch` leaves behind -- at exactly that step
    # (scratch/step_jitter.py, scratch/trace_stall.sh: 36 ms once, never with gc.freeze()).  It is the Python veneer's
    # garbage collector, not the runtime and not the library; a Rust host has none.  So: collect once, freeze.)
    import gc

    gc.collect()
    gc.freeze()
    t_settle = time.perf_counter()
    if dist:
        # every step holds a collective: all ranks must run the SAME number of settle steps, so the count cannot
        # depend on a local clock
        # (from the shape, identical on every rank: ~6.5 TB/s over the shard + 0.1 ms of fixed cost per step)
        est_ms = (args.rows / world) * args.dim * (2 if args.dtype == "f16" else 4) / 6.5e9 + 0.1
        for _ in range(max(256, int(args.settle_ms / est_ms)) if args.settle_ms > 0 else 0):
            step(0)
    else:
        while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            step(0)
    for i in range(args.warmup):
        step(i)

    ix.profile_read(reset=True)
    ix.profile_enable(not args.no_profile)  # HIP events around each stage, on the stream the kernels run on
    sh.time_exchange(use_sharded and not args.no_profile)  # torch events around all-gather + merge, on torch's stream
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ix.profile_enable(False)
    prof = ix.profile_read()
    exchange_ms = sh.exchange_ms_per_step()
    sh.time_exchange(False)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    elem = 2 if args.dtype == "f16" else 4
    scan_ms = prof.scan_ms / max(prof.n_scan_launches, 1)
    image_scan = args.image_scan and args.dtype == "f32"
    q8_scan = args.q8_scan and args.dtype == "f32"
    if image_scan:
        elem = 2  # the nomination scan streams the binary16 image: those are the bytes this kernel has to read
    if q8_scan:
        elem = 1  # one byte per element (+ 4 B of scale per row, not counted)
    bytes_per_launch = n_local * args.dim * elem
    achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    batched = args.batch > 1
    out = {
        "metric": "queries/sec, 768-d cosine top-100 over 10M chunks (single query, f32)" if not batched else
                  f"queries/sec, {args.dim}-d cosine top-{args.k} over {args.rows} chunks ({args.batch} batched queries)",
        "value": args.steps * args.batch / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"{args.rows} chunks x {args.dim}-d {args.dtype}, {args.batch} query/step, top_k={args.k}, "
                               f"corpus row-sharded over {world} GPU(s), exact scan + re-score",
                   "rows_per_gpu": n_local, "parallelism": f"row-shard x{world}" if world > 1 else "single GPU"},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "traffic_source": None,
            "kernel": "scan_fixed_kernel",
            "bytes_per_launch": bytes_per_launch,
            "kernel_ms": scan_ms,
        },
        "stages_ms": {"scan": scan_ms, "select": prof.select_ms / max(prof.n_scan_launches, 1),
                      "rescore_sort": prof.rescore_ms / max(prof.n_scan_launches, 1)},
        "candidates_per_query": prof.n_candidates / max(prof.n_searches, 1),
        "band_retries": prof.n_retries,
        "fill_s": round(fill_s, 2),
        "build_source_sha16": source_sha16(),
    }
    if measured and "error" not in measured:
        out["roofline"]["peak_measured_read_GBps"] = measured["read"]
        out["roofline"]["peak_measured_copy_GBps"] = measured["copy"]
        out["roofline"]["frac_of_measured"] = achieved / measured["read"] if measured["read"] > 0 else None
        out["roofline"]["measured_note"] = ("rlr_index_probe_bandwidth over this index's rows before the timed region: read = the "
                                            "scan's stream without arithmetic (best of three launch shapes), copy = hipMemcpyAsync "
                                            "device-to-device, bytes read + written; `frac` stays achieved / the 8 TB/s spec peak")
    elif measured:
        out["roofline"]["measured_error"] = measured["error"]
    if use_sharded:
        # the exchange step of SURVEY 8(e): all-gather of world x k packed results + merge kernel, per step
        out["stages_ms"]["allgather_merge"] = exchange_ms
    if batched and prof.n_batches:
        out["roofline"] = batched_roofline(prof, args.dim, args.batch, args.image) or out["roofline"]
        out["stages_ms"] = {"gemm_all_launches": prof.batch_gemm_ms / prof.n_batches,
                            "gemm_main_pass": prof.batch_main_ms / prof.n_batches,
                            "select_and_finish": prof.batch_other_ms / prof.n_batches}
        out["band_retries"] = prof.n_batch_fallbacks
        out["dtype"] = f"{args.dtype} rows, f16 MFMA nomination + f32 reference-order re-score"
    if q8_scan and not batched:
        out["roofline"]["kernel"] = q8_kernel_name(args.dim)
        out["dtype"] = "f32 rows, 8-bit nomination scan + f32 reference-order re-score"
        out["config"]["workload"] += "; single-query nomination over the 8-bit copy (opt-in, +dim+4 B/row of HBM)"
    elif image_scan and not batched:
        out["roofline"]["kernel"] = "scan_image_kernel"
        out["dtype"] = "f32 rows, binary16 nomination scan over the image + f32 reference-order re-score"
        out["config"]["workload"] += "; single-query nomination over the binary16 image (opt-in, +dim*2 B/row of HBM)"
    if not batched:
        quote_traffic(out["roofline"], pmc_traffic(bytes_per_launch, out["roofline"]["kernel"]), "rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE")
    extras = world == 1 and not batched and not args.image_scan and not args.q8_scan and not args.no_extras
    # Informational, outside the timed region above: the same workload with the opt-in nomination copies (identical
    # results, the scan streams a half / a quarter of the bytes).  Never the headline `value`.
    if extras and args.dtype == "f32" and args.dim % 64 == 0 and n_local * args.dim * 3 < 100e9:
        out["optional_modes"] = {}
        modes = [("image_scan", dict(on=True, single_query=True), 2, "scan_image_kernel",
                  "single queries nominate over the binary16 image (rlr_index_enable_batch_image(idx, 3)); "
                  "+dim*2 B/row of HBM, results identical")]
        if args.dim <= 2048:
            modes.append(("q8_scan", dict(on=False, q8=True), 1, q8_kernel_name(args.dim),
                          "single queries nominate over the 8-bit copy with per-row scales "
                          "(rlr_index_enable_batch_image(idx, 4)); +dim+4 B/row of HBM, results identical"))
        for name, kw, eb, kern, what in modes:
            try:
                ix.enable_batch_image(**kw)
                for i in range(20):
                    step(i % n_q)
                ix.profile_read(reset=True)
                ix.profile_enable(True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    alt = step((args.warmup + i) % n_q)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                ix.profile_enable(False)
                p2 = ix.profile_read()
                k_ms = p2.scan_ms / max(p2.n_scan_launches, 1)
                same = bool(last is not None and np.array_equal(alt[0], last[0]) and
                            np.array_equal(alt[1].view(np.uint32), last[1].view(np.uint32)))
                out["optional_modes"][name] = {
                    "what": what, "value": args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3,
                    "kernel": kern, "kernel_ms": k_ms, "bytes_per_launch": n_local * args.dim * eb,
                    "achieved_GBps": n_local * args.dim * eb / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                    "candidates_per_query": p2.n_candidates / max(p2.n_searches, 1),
                    "same_result_as_the_f32_scan_on_the_last_query": same}
            except Exception as e:  # never let the extra measurement take the headline line down
                out["optional_modes"][name] = {"error": str(e)}
            ix.enable_batch_image(False)
    # BASELINE's other single-GPU configurations, each with its own roofline, outside the timed region too
    if extras and args.dtype == "f32" and args.dim == 768 and args.rows == 10_000_000:
        out["configs"] = {}
        headline_ms = elapsed / args.steps * 1e3
        tight = 0x80000000
        for name, fn in (("headline_shard_1of8", lambda: config_shard_1of8(rlr, torch, sharded, args, headline_ms)),
                         ("C3_256_batched_queries", lambda: config_c3(rlr, ix, args, torch)),
                         ("C2_100k_mmr", lambda: config_c2(rlr, torch)),
                         ("C4_per_gpu_share", lambda: config_single(
                             rlr, torch, args, 12_500_000, args.seed + 4, 0,
                             "C4 per-GPU share: 12500000 chunks x 768-d f32 (100 M / 8), 1 query/step, top_k=100")),
                         ("headline_clustered", lambda: config_single(
                             rlr, torch, args, args.rows, args.seed + 6, 64 | tight,
                             f"{args.rows} chunks x 768-d f32 in 64 TIGHT clusters (rows of a cluster are near-copies, cosine "
                             f"~0.999: the top of every ranking is one dense cluster), 1 query/step, top_k=100")),
                         ("headline_duplicates", lambda: config_single(
                             rlr, torch, args, args.rows, args.seed, 0x40000000,
                             f"{args.rows} chunks x 768-d f32, the last 1 % of the rows exact duplicates of the first 1 % "
                             f"(re-ingested documents, rag_engine.rs:347-384), 1 query/step, top_k=100"))):
            try:
                out["configs"][name] = fn()
            except Exception as e:
                out["configs"][name] = {"error": str(e)}
        for name in ("headline_clustered", "headline_duplicates"):
            c = out["configs"].get(name, {})
            if "value" in c:
                c["vs_iid_headline"] = c["value"] / out["value"]
    if world == 1 and not args.no_cpu and O is not None:
        base, sample_rows, want = cpu_baseline(args, rlr)
        out["cpu_baseline"] = base
        # parity spot-check on the sample: the GPU over the same rows must agree bit for bit
        if args.check:
            with rlr.GpuIndex(args.dim, args.dtype) as chk:
                chk.fill_synthetic(sample_rows.shape[0], args.seed)
                r, c = chk.search_topk(rlr.normalize(O.synth_query(args.dim, args.seed + 1)), args.k)
                ok = bool(np.array_equal(r[0], want[0]) and
                          np.array_equal(c[0].view(np.uint32), want[2].view(np.uint32)))
            out["parity_check"] = {"rows": int(sample_rows.shape[0]), "top_k_identical_and_scores_bit_equal": ok}
        del sample_rows
    else:
        out["cpu_baseline"] = None
    if "configs" in out:
        # the C5 share needs 12.8 GB of rows + 12.8 GB of image: after the headline corpus is gone
        sh.index.close()
        try:
            out["configs"]["C5_per_gpu_share"] = config_c5_share(rlr, torch)
        except Exception as e:
            out["configs"]["C5_per_gpu_share"] = {"error": str(e)}
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
