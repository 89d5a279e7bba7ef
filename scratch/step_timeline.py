"""Kernel timeline of the median step from a rocprofv3 kernel trace of a one-query-per-step loop: steps are split at the scan
kernel; prints every kernel of the median step (start / end relative to the scan's start), the device-idle gap to the next
step's first kernel, and the mean over all steps.   python scratch/step_timeline.py <kernel_trace.csv> [min_steps]"""
import csv, re, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:30]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
scan_idx = [i for i, k in enumerate(ks) if k[0].startswith("scan_")]
steps = []
for a, b in zip(scan_idx, scan_idx[1:]):
    # a step = [copy in front of the scan] scan ... up to the copy in front of the next scan
    up = ("__amd_rocclr_copy", "stage_query_kernel")  # the query upload in front of a scan
    lo = a - 1 if a > 0 and ks[a - 1][0].startswith(up) else a
    hi = b - 1 if ks[b - 1][0].startswith(up) else b
    steps.append(ks[lo:hi] + [("NEXT", ks[hi][1], ks[hi][1])])
steps = [s for s in steps if len(s) == statistics.mode(len(x) for x in steps)][20:]
def total(s): return (s[-1][1] - s[0][1]) / 1000
tot = sorted(total(s) for s in steps)
print("steps %d  step period us: median %.1f min %.1f mean %.1f" % (len(steps), statistics.median(tot), tot[0], statistics.mean(tot)))
s = sorted(steps, key=total)[len(steps) // 2]
t0 = s[0][1]
prev_end = None
for k in s:
    gap = "" if prev_end is None else " gap %5.1f" % ((k[1] - prev_end) / 1000)
    print(f"{k[0]:34s} start {(k[1]-t0)/1000:8.1f} end {(k[2]-t0)/1000:8.1f} dur {(k[2]-k[1])/1000:7.1f}{gap}")
    prev_end = k[2]
names = [k[0] for k in s[:-1]]
for j, nme in enumerate(names):
    print("  mean %-32s %.2f us" % (nme, statistics.mean((st[j][2] - st[j][1]) / 1000 for st in steps)))
print("  mean idle before next step %.2f us" % statistics.mean((st[-1][1] - st[-2][2]) / 1000 for st in steps))
