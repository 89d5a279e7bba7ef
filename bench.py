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

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


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
    ap.add_argument("--settle-ms", type=float, default=500.0,
                    help="untimed clock/power settle phase before the warmup steps (0 disables)")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational optional-mode measurement")
    ap.add_argument("--image-scan", action="store_true",
                    help="single queries nominate over the binary16 image too (half the scan bytes; same results)")
    ap.add_argument("--q8-scan", action="store_true",
                    help="single queries nominate over the 8-bit copy (a quarter of the scan bytes; same results)")
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


def pmc_traffic(bytes_per_launch, kernel):
    """HBM bytes per launch of the scan kernel from the committed rocprofv3 --pmc passes of
    this same command (profiles/rNN_pmc.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    separate passes).  PMC counters cannot be read from inside the process, so the figure is
    quoted from the newest summary of the SAME kernel whose byte count matches this run's shape
    (within 5 %); any other shape -- a shard of an N > 1 run, other dims -- reports null."""
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
            best = (t, os.path.basename(f))
    return best


def cpu_baseline(args, rlr):
    """Times the oracle (CPU port of the reference loops) on a bounded sample of the same
    corpus: rows [0, cpu_rows) of the synthetic stream, full search (scan + stable sort +
    take), 1 thread -- the reference's actual behaviour -- then the same arithmetic with
    rows split over all host cores.  Extrapolated linearly to the full row count."""
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
        O.scan(rows, qn, threads=cores)
    tm = (time.perf_counter() - t0) / reps
    scale = args.rows / n
    return {
        "value": 1.0 / (t1 * scale),
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"oracle search (scan + stable sort + take) of 1 query over rows [0,{n}) of the same synthetic "
                   f"corpus, {t1 * 1e3:.0f} ms/query on 1 thread, extrapolated x{scale:.0f} to {args.rows} rows; "
                   f"contiguous matrix, no per-candidate clone (both favour the reference)"),
        "all_cores": {"value": 1.0 / (tm * scale), "cores": cores,
                      "note": "scan only, rows split over threads; not something the reference does"},
        "sample_gen_s": round(gen_s, 2),
    }, rows, r1


def ensure_built():
    """librlr_gpu.so normally travels with the tree; if it does not, compile it once (hipcc is a child
    process, nothing here touches the GPU) -- one rank builds, the others wait on the lock."""
    so = os.path.join(ROOT, "rust-local-rag_amd", "librlr_gpu.so")
    if os.path.exists(so):
        return
    import fcntl

    with open(os.path.join(ROOT, "rust-local-rag_amd", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so):
            import __graft_entry__

            __graft_entry__._load_build_module().build()


def main():
    args = parse()
    ensure_built()
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

    force_sharded = os.environ.get("RLR_BENCH_FORCE_SHARDED") == "1" or force_dist  # rehearse the N>1 code path on one GPU

    def step(i):
        if world == 1 and not force_sharded:
            return ix.search_topk(qs[i], args.k)
        return sh.search_topk(qs[i], args.k)

    # Untimed settle phase before the W warmup steps.  Two things are kept out of the timed region:
    # (1) a cold GPU needs some hundred ms of work before clocks / HBM power state are at their sustained
    # level; (2) the HIP runtime grows its per-queue launch resources once, about 200 searches (~1000
    # kernel launches) into a process, which stalls that one call for 30-40 ms (measured with
    # scratch/step_jitter.py: call 206 at every corpus size, never again afterwards).
    t_settle, n_settle = time.perf_counter(), 0
    if dist:
        # every step holds a collective: all ranks must run the SAME number of settle steps, so the count cannot
        # depend on a local clock
        for _ in range(768 if args.settle_ms > 0 else 0):
            step(0)
    else:
        while args.settle_ms > 0 and ((time.perf_counter() - t_settle) * 1e3 < args.settle_ms or n_settle < 256):
            step(0)
            n_settle += 1
    for i in range(args.warmup):
        step(i)

    ix.profile_read(reset=True)
    ix.profile_enable(not args.no_profile)  # HIP events around each stage, on the stream the kernels run on
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
    }
    if batched and prof.n_batches:
        gemm_ms = prof.batch_gemm_ms / prof.n_batches
        passes = (args.batch + 255) // 256
        b_bytes = passes * bytes_per_launch
        out["roofline"] = {
            "bound": "hbm", "achieved": b_bytes / (gemm_ms * 1e-3) / 1e9 if gemm_ms > 0 else 0.0, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": (b_bytes / (gemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if gemm_ms > 0 else 0.0,
            "traffic": None, "kernel": "gemm_nominate_kernel (both launches of a batch: sample + filter)",
            "bytes_per_launch": b_bytes, "kernel_ms": gemm_ms,
            "mfma": {"achieved_tflops": prof.batch_gemm_flops / prof.n_batches / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                     "peak_tflops_f16_dense": 2500.0, "note": "f16 MFMA nominates; results re-scored in f32 reference order"},
        }
        out["stages_ms"] = {"gemm": gemm_ms, "select_and_finish": prof.batch_other_ms / prof.n_batches}
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
    t = pmc_traffic(bytes_per_launch, out["roofline"]["kernel"]) if not batched else None
    if t:
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = t[0], f"profiles/{t[1]} (rocprofv3 --pmc)"
    # Informational, outside the timed region above: the same workload with the opt-in nomination copies (identical
    # results, the scan streams a half / a quarter of the bytes).  Never the headline `value`.
    if (world == 1 and not batched and not args.image_scan and not args.q8_scan and not args.no_extras
            and args.dtype == "f32" and args.dim % 64 == 0 and n_local * args.dim * 3 < 100e9):
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
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
