"""Sum one rocprofv3 PMC counter per kernel name (counter_collection CSV)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2]
tot = collections.defaultdict(float)
cnt = collections.Counter()
for r in rows:
    if r.get("Counter_Name") != want:
        continue
    n = re.sub(r"itts::\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
    tot[n] += float(r["Counter_Value"])
    cnt[n] += 1
print(f"counter {want}: per-kernel totals (raw counter units; FETCH_SIZE/WRITE_SIZE are KiB, FETCH_SIZE under-reports 2x on gfx950)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{k:62s} n={cnt[k]:6d} total={v:14.1f} per_launch={v / cnt[k]:12.2f}")
dec = sum(v for k, v in tot.items() if "gemv" in k or "decode_attn" in k or "sampler" in k)
ndec = max(cnt.get(next((k for k in cnt if "sampler" in k), ""), 1) - 1, 1)
print(f"decode-step kernels total={dec:.1f} over {ndec} steps -> per step {dec / ndec:.1f}")
