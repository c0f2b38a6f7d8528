"""Idle time on the GPU timeline outside the decode graphs: for a rocprofv3 kernel trace of bench.py, sum the gaps that follow
each kernel family (host launch rate, not the GPU, bounds phases made of many short launches)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
decode = ("gemv_bf16", "decode_attn2", "sampler")
busy = idle = 0
gaps = collections.Counter()
n = collections.Counter()
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    d0 = any(x in k0 for x in decode)
    d1 = any(x in k1 for x in decode)
    if d0 and d1:
        continue
    busy += e0 - s0
    g = max(0, s1 - e0)
    if g > 200000:  # > 200 us: step boundary / host work, not launch latency
        continue
    idle += g
    key = k0.split("<")[0].split("(")[0][-40:]
    gaps[key] += g
    n[key] += 1
print(f"outside decode: busy {busy/1e6:.2f} ms, idle between launches {idle/1e6:.2f} ms")
for k, v in gaps.most_common(12):
    print(f"  after {k:42s} n={n[k]:5d}  idle {v/1e6:7.3f} ms  ({v/max(n[k],1)/1e3:.1f} us each)")
pairs = collections.Counter()
pgap = collections.Counter()
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    if "copyBuffer" in k0 or "copyBuffer" in k1:
        a = k0.split("<")[0].split("(")[0][-28:]
        b = k1.split("<")[0].split("(")[0][-28:]
        pairs[(a, b)] += 1
        pgap[(a, b)] += max(0, s1 - e0)
print("copyBuffer neighbours (prev -> next): count, total gap ms")
for k, v in pairs.most_common(12):
    print(f"  {k[0]:30s} -> {k[1]:30s} n={v:5d} gap {pgap[k]/1e6:8.3f} ms")
