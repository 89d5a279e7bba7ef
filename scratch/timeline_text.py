"""Timeline of one fused text search from a rocprofv3 kernel trace of scratch/time_c2_text_abi.py: every kernel of the median call
(by span), start / end relative to the scan's start, with its queue.   python scratch/timeline_text.py <kernel_trace.csv>"""
import csv, re, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:30]
calls, cur = [], []
for r in rows:
    n = short(r["Kernel_Name"])
    if n == "scan_fixed_kernel" and cur:
        calls.append(cur); cur = []
    cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id")))
calls.append(cur)
div = [c for c in calls if any(k[0].startswith("mmr_greedy") for k in c) and any(k[0] == "bm25_terms_kernel" for k in c)]
def span(c):
    ks = [k for k in c if not k[0].startswith("__amd_rocclr_copy")]
    return (max(k[2] for k in ks) - min(k[1] for k in ks)) / 1000
spans = sorted(span(c) for c in div)
print("calls", len(div), "kernel span us: median %.1f min %.1f" % (statistics.median(spans), spans[0]))
c = sorted(div, key=span)[len(div) // 2]
t0 = min(k[1] for k in c)
for k in sorted(c, key=lambda k: k[1]):
    print(f"{k[0]:32s} q{k[3]} start {(k[1]-t0)/1000:7.1f} end {(k[2]-t0)/1000:7.1f} dur {(k[2]-k[1])/1000:6.1f}")
