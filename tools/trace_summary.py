"""Summarise a rocprofv3 kernel-trace CSV: per (kernel, grid) durations and inter-kernel gaps."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))


def short(n):
    n = re.sub(r"itts::\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN4itts12_GLOBAL__N_1\d+", "", n)
    return n[:70]


d = collections.defaultdict(list)
for r in rows:
    gx = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    wx = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    d[(short(r["Kernel_Name"]), gx, wx)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print("total kernel time %.2f ms over %d dispatches" % (tot / 1e6, len(rows)))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:28]:
    v.sort()
    print("%-72s grid %-8s wg %-5s n=%-6d avg %8.1f us  med %8.1f  min %8.1f  tot %8.2f ms (%4.1f%%)" % (
        k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
ds = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
gaps = [b[0] - a[1] for a, b in zip(ds, ds[1:]) if ("gemv" in a[2] or "attn2" in a[2]) and ("gemv" in b[2] or "attn2" in b[2])]
if gaps:
    gaps.sort()
    print("decode inter-kernel gaps: n=%d med %.2f us avg %.2f us p90 %.2f us" % (
        len(gaps), gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
