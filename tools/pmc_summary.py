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
# decode-step kernels only: gemv / cache attention / sampler launches of the per-token loop.  The prefill's head gemv
# and sampler (one launch each) are excluded by counting per kernel: launches_per_step * steps.
# (greedy steps of the persistent engine have no sampler launch - the sampler runs inside decode_engine_kernel: one launch a step)
neng = max((c for k, c in cnt.items() if "decode_engine" in k), default=0)
nsamp = max((c for k, c in cnt.items() if "sampler" in k), default=1)
steps = neng if neng > nsamp else max(nsamp - 1, 1)
dec = 0.0
for k, v in tot.items():
    if "gemv" in k or "decode_attn" in k or "sampler" in k or "decode_engine" in k or "skinny_mfma" in k or "ln_rows" in k:
        per = v / cnt[k]
        launches_in_loop = cnt[k] - (1 if cnt[k] % steps == 1 else 0)
        dec += per * launches_in_loop
print(f"decode-step kernels: {dec / steps:.1f} counter units per step over {steps} steps")
if len(sys.argv) > 3:
    import json

    json.dump({"counter": want, "steps": steps, "per_step_units": dec / steps}, open(sys.argv[3], "w"))
