import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["steps"], d["warmup"], d["ms_per_step"])
for k,v in d["also"].items(): print(k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","vs_headline","decode_ms_per_token_step","error","decode_mode")})
