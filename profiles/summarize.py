#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<tag>_*) into the small summaries kept under profiles/.

  headline scan (kernel trace + the two HBM counter passes):
      python profiles/summarize.py <tag> <kt dir> <fetch dir> <write dir> "<command>" [kernel substring]
      -> <tag>_kernel_stats.csv (verbatim --stats table), <tag>_pmc.json (HBM traffic per launch)

  any other run: the --stats table plus one JSON with the named kernels' averages, optional SQ / HBM counter passes:
      python profiles/summarize.py kernels <out tag> <kt dir> "<command>" <substr,substr,...> [--pmc dir ...]
      (counters are reported for the FIRST substring's kernel only)
      -> <out tag>_kernel_stats.csv, <out tag>_kernels.json
      The --stats table lumps every dispatch of a kernel into one row, but a batch may launch the same GEMM kernel more than
      once with the SAME persistent grid (until late in round 3: over a short row range -- the "bootstrap" -- and over the rest
      of the corpus, the main pass the roofline claims are about; RLR_BATCH_RANK_DIV=1 still does).  So the per-dispatch trace (<kt dir>/*_kernel_trace.csv) is also read: it is cut into batches at every
      dispatch of the anchor kernel (prep_queries_kernel: the first launch of a batch) and the FIRST substring's kernel is
      reported per ordinal inside its batch ("dispatch_groups": calls / avg / min / max of the 1st, 2nd, ... launch per
      batch); the counter passes are grouped the same way ("counters_by_ordinal").

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B/4-B
per-lane streaming stores.  SQ_* counters are reported as the mean per launch of the kernel they are filtered to
(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def source_sha16():
    """sha256 over the sources librlr_gpu.so is built from (csrc/ + include/), first 16 hex digits -- bench.py computes the same
    and quotes a summary's counter traffic only for the build it was collected on"""
    import hashlib

    root = os.path.dirname(HERE)
    h = hashlib.sha256()
    for d in (os.path.join(root, "rust-local-rag_amd", "csrc"), os.path.join(root, "include")):
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p):
                h.update(f.encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def counter_means(d, kernel_substr):
    """{counter: (mean per dispatch, dispatches)} of the kernel (by substring) with the most dispatches in `d`"""
    fs = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        return "", {}
    fs = fs[-1:]  # a re-used output directory keeps older runs' files: take the newest
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if kernel_substr in r["Kernel_Name"]:
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        return "", {}
    name = max(vals, key=lambda k: max(len(v) for v in vals[k].values()))
    return name, {c: (sum(v) / len(v), len(v), max(v)) for c, v in vals[name].items()}


def stats_rows(d_kt):
    fs = sorted(glob.glob(os.path.join(d_kt, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit(f"no kernel_stats.csv under {d_kt}")
    return fs[-1], list(csv.DictReader(open(fs[-1])))  # newest: a re-used directory keeps older runs' files


def ordinal_groups(rows, kernel_substr, anchor="prep_queries_kernel", name_key="Kernel_Name"):
    """rows: per-dispatch records in dispatch order -> {kernel name: {ordinal within its batch: [records]}}; a batch starts at
    every dispatch of `anchor` (no anchor in the trace: the whole run is one batch and the ordinal is the running count)"""
    groups = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.Counter()
    for r in rows:
        name = r[name_key]
        if anchor in name:
            seen.clear()
        if kernel_substr in name:
            seen[name] += 1
            groups[name][seen[name]].append(r)
    return groups


def trace_rows(d_kt):
    fs = sorted(glob.glob(os.path.join(d_kt, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        return []
    rows = list(csv.DictReader(open(fs[-1])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def short_name(name):
    key = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rlr::", "").split("(")[0]
    if key.startswith("_Z"):  # a mangled name the tool left as it was: keep the readable middle
        key = next((w for w in ("prep_queries_kernel", "build_image_kernel", "gemm_resident_kernel") if w in key), key)
    return key


def headline():
    tag, d_kt, d_fetch, d_write, command = sys.argv[1:6]
    kernel = sys.argv[6] if len(sys.argv) > 6 else "scan_"
    stats, rows = stats_rows(d_kt)
    shutil.copy(stats, os.path.join(HERE, f"{tag}_kernel_stats.csv"))
    dom = next(r for r in rows if kernel in r["Name"])
    name, f = counter_means(d_fetch, kernel)
    _, w = counter_means(d_write, kernel)
    fetch_kib, n_f, _ = f["FETCH_SIZE"]
    write_kib, n_w, _ = w["WRITE_SIZE"]
    out = {
        "command": command,
        "build_source_sha16": source_sha16(),
        "kernel": name,
        "kernel_trace": {"calls": int(dom["Calls"]), "avg_ns": float(dom["AverageNs"]),
                         "min_ns": float(dom["MinNs"]), "max_ns": float(dom["MaxNs"]),
                         "pct_of_gpu_time": float(dom["Percentage"])},
        "pmc": {"FETCH_SIZE_KiB_raw_mean": fetch_kib, "fetch_dispatches": n_f,
                "WRITE_SIZE_KiB_mean": write_kib, "write_dispatches": n_w,
                "fetch_bytes_corrected_x2": 2.0 * fetch_kib * 1024.0,
                "write_bytes": write_kib * 1024.0,
                "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0},
    }
    with open(os.path.join(HERE, f"{tag}_pmc.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


def kernels():
    tag, d_kt, command, subs = sys.argv[2:6]
    pmc_dirs = sys.argv[7:] if len(sys.argv) > 6 and sys.argv[6] == "--pmc" else []
    stats, rows = stats_rows(d_kt)
    shutil.copy(stats, os.path.join(HERE, f"{tag}_kernel_stats.csv"))
    out = {"command": command, "build_source_sha16": source_sha16(), "kernels": {}}
    for sub in subs.split(","):
        hit = [r for r in rows if sub in r["Name"]]
        for r in hit:
            entry = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                     "max_us": float(r["MaxNs"]) / 1e3, "pct_of_gpu_time": float(r["Percentage"])}
            out["kernels"][short_name(r["Name"])] = entry
        if pmc_dirs and sub == subs.split(",")[0]:
            # counters of the FIRST kernel of the list only (the dominant one); `max` = its largest launch (a batch
            # launches the same kernel over a small and a large row range: the large one is the one the claims are about)
            counters = {}
            for d in pmc_dirs:
                kname, c = counter_means(d, sub)
                for k, (mean, n, mx) in c.items():
                    counters[k] = {"mean_per_launch": mean, "max_launch": mx, "launches": n}
            if "FETCH_SIZE" in counters:
                counters["FETCH_SIZE"]["max_launch_bytes_corrected_x2"] = 2.0 * 1024.0 * counters["FETCH_SIZE"]["max_launch"]
            if counters:
                out["counters"] = {"kernel": kname, "values": counters}
            # the same kernel per ordinal inside a batch: durations from the per-dispatch trace, counters from the passes
            dg = {}
            for name, by_ord in ordinal_groups(trace_rows(d_kt), sub).items():
                lst = []
                for o in sorted(by_ord):
                    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in by_ord[o]]
                    lst.append({"ordinal_in_batch": o, "calls": len(us), "avg_us": sum(us) / len(us), "min_us": min(us), "max_us": max(us)})
                dg[short_name(name)] = lst
            if dg:
                out["dispatch_groups"] = dg
            cbo = {}
            for d in pmc_dirs:
                fs = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
                if not fs:
                    continue
                recs = list(csv.DictReader(open(fs[-1])))
                # one record per (dispatch, counter): walk the dispatches in order, once per counter
                for cname in sorted({r["Counter_Name"] for r in recs}):
                    one = sorted((r for r in recs if r["Counter_Name"] == cname), key=lambda r: int(r["Dispatch_Id"]))
                    for name, by_ord in ordinal_groups(one, sub).items():
                        for o in sorted(by_ord):
                            v = [float(r["Counter_Value"]) for r in by_ord[o]]
                            e = cbo.setdefault(short_name(name), {}).setdefault(str(o), {})
                            e[cname] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
                            if cname == "FETCH_SIZE":
                                e[cname]["bytes_corrected_x2"] = 2.0 * 1024.0 * sum(v) / len(v)
                            if cname == "WRITE_SIZE":
                                e[cname]["bytes"] = 1024.0 * sum(v) / len(v)
            if cbo:
                out["counters_by_ordinal"] = cbo
    with open(os.path.join(HERE, f"{tag}_kernels.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "kernels":
        kernels()
    else:
        headline()
