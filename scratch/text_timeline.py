"""Kernel timeline of the text search (two streams: scan chain + BM25 chain) from a rocprofv3 kernel trace of
time_c2_hybrid.py hybrid-only: steps split at the scan kernel, only steps that end in the greedy MMR kernel; prints the
median-length step with each kernel's start / end relative to the scan start and its stream (queue id).
  python scratch/text_timeline.py <kernel_trace.csv>"""
import csv, re, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:30]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
scan_idx = [i for i, k in enumerate(ks) if k[0].startswith("scan_")]
steps = [ks[a:b] for a, b in zip(scan_idx, scan_idx[1:])]
want = sys.argv[2] if len(sys.argv) > 2 else "mmr_greedy"
steps = [s for s in steps if any(k[0].startswith(want) for k in s)]
mode = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == mode][20:]
def span(s): return (max(k[2] for k in s) - s[0][1]) / 1000
sp = sorted(span(s) for s in steps)
print("steps %d  device span us: median %.1f min %.1f mean %.1f" % (len(steps), statistics.median(sp), sp[0], statistics.mean(sp)))
s = sorted(steps, key=span)[len(steps) // 2]
t0 = s[0][1]
last_end = {}
for k in s:
    g = last_end.get(k[3])
    gap = "" if g is None else " gap(q) %5.1f" % ((k[1] - g) / 1000)
    print(f"q{k[3]:>3s} {k[0]:30s} start {(k[1]-t0)/1000:7.1f} end {(k[2]-t0)/1000:7.1f} dur {(k[2]-k[1])/1000:6.1f}{gap}")
    last_end[k[3]] = k[2]
names = [k[0] for k in s]
for j, nme in enumerate(names):
    same = [st for st in steps if st[j][0] == nme]
    print("  mean %-30s dur %.2f  start %.1f" % (nme, statistics.mean((st[j][2] - st[j][1]) / 1000 for st in same),
                                               statistics.mean((st[j][1] - st[0][1]) / 1000 for st in same)))
