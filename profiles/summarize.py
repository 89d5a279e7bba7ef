#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof_*) into the small summaries kept
under profiles/:  <tag>_kernel_stats.csv (verbatim --stats table) and <tag>_pmc.json
(HBM traffic per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is
doubled; WRITE_SIZE is exact for 16-B/4-B per-lane streaming stores.

    python profiles/summarize.py r01 gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write "<command>"
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counter_mean(d, counter, kernel_substr):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max(vals, key=lambda k: len(vals[k]))
    v = vals[name]
    return name, sum(v) / len(v), len(v)


def main():
    tag, d_kt, d_fetch, d_write, command = sys.argv[1:6]
    kernel = sys.argv[6] if len(sys.argv) > 6 else "scan_"
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(d_kt, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    dom = next(r for r in rows if kernel in r["Name"])
    name, fetch_kib, n_f = counter_mean(d_fetch, "FETCH_SIZE", kernel)
    _, write_kib, n_w = counter_mean(d_write, "WRITE_SIZE", kernel)
    out = {
        "command": command,
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
    with open(os.path.join(here, f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
