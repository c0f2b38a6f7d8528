"""Per-kernel averages of every collected counter (rocprofv3 counter_collection CSV) for the vocoder kernels, with the
kernel's average duration from the kernel trace of the same run."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[(re.sub(r"itts::\(anonymous namespace\)::", "", r["Kernel_Name"])[:48], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
except Exception as e:  # noqa: BLE001
    print("no kernel trace:", e)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    n = re.sub(r"itts::\(anonymous namespace\)::", "", r["Kernel_Name"])[:48]
    if not any(k in n for k in ("snake", "conv_lds", "gemm_glds", "gemm_mfma")):
        continue
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
print("per-launch averages (counter units as rocprofv3 reports them: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves or SIMDs)")
print(f"{'kernel':50s} {'grid':>9s} {'n':>4s} {'us':>8s} " + " ".join(f"{c[-18:]:>18s}" for c in names))
for key in sorted(tot, key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(key, [])
    n = max(cnt[key].values())
    print(f"{key[0]:50s} {key[1]:>9s} {n:4d} {(sum(d) / len(d) if d else 0):8.1f} " + " ".join(f"{tot[key][c] / max(cnt[key][c], 1):18.0f}" for c in names))
