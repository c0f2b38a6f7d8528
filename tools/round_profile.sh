#!/bin/bash
# Round profile: (1) bench with cpu baseline, (2) rocprofv3 --kernel-trace --stats of the same bench command,
# (3) PMC passes (FETCH_SIZE, WRITE_SIZE) on a short decode for the HBM-traffic column.  Run on the GPU box:
#   tools/round_profile.sh r01
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ -z "$ONLY_PMC" ]; then
python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $out/bench_profiled.json 2> $out/prof.err
t=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats
head -12 $out/kernel_summary.txt | cut -c1-200
fi
# PMC passes: eager launches (counters hang under hipGraph replay); each counter in its own pass (TCC slot limits),
# --kernel-trace only beside --pmc.  With the persistent decode engine a token step is 3 dispatches, so the passes run the
# bench's own workload at full length (T = 480, L = 105: 1.4 k engine launches).  (r01 / r02 ran the 122-launches-per-step
# path, 58 k dispatches at T = 480, where the profiled process died with SIGSEGV - see profiles/README.md - and so used 48
# steps at the bench's mean sequence length: PMC_T=48 PMC_L=320 ITTS_ENGINE=0 reproduces those passes.)
PT=${PMC_T:-480}
PL=${PMC_L:-105}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --mel-tokens $PT --text-tokens $PL --no-graph --no-cpu-baseline --no-also > $out/pmc_$c.json 2> $out/pmc_$c.err
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" $c $out/pmc_${c}_step.json > $out/pmc_$c.txt
  rm -rf $out/pmc_$c
  head -14 $out/pmc_$c.txt | cut -c1-170
done
python3 - <<PY
import json
f = json.load(open("$out/pmc_FETCH_SIZE_step.json")); w = json.load(open("$out/pmc_WRITE_SIZE_step.json"))
D, NL, V, B, L, T = 1280, 24, 8194, 2, $PL, $PT
alg = (NL * (12 * D * D + 13 * D) + 4 * D + D * V + V) * 2 + B * 2 * NL * D * 2 * ((32 + L + 2 + 1) + T / 2.0)
hbm = (2.0 * f["per_step_units"] + w["per_step_units"]) * 1024   # FETCH_SIZE under-reports 2x on gfx950 (MI355X_MICROARCH HBM)
import subprocess
head = subprocess.run(["git", "-C", "$GRAFT_REPO_ROOT", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "$(cat $GRAFT_REPO_ROOT/.git_head 2>/dev/null)"
json.dump({"kernels_head": head, "engine": "$(echo ${ITTS_ENGINE:-1})", "mel_tokens": T, "text_tokens": L, "mean_S": (32 + L + 2 + 1) + T / 2.0, "decode_rows": B, "fetch_kib_raw_per_step": f["per_step_units"], "write_kib_per_step": w["per_step_units"],
           "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": hbm / alg},
          open("$out/pmc_decode.json", "w"), indent=1)
print(open("$out/pmc_decode.json").read())
PY
# BASELINE config 3's decode step (64 rows: 97 skinny MFMA projections + 49 LayerNorm + 24 cache attention + sampler per step): the
# same two passes on `--batch 32`, 24 steps behind a text prefix lengthened so that the mean sequence length is the timed run's
# (S = 32 + 332 + 3 + 12 = 379 against 380: same kernels, same grids, same bytes per step; ~4 k counted dispatches).  A counted
# dispatch takes tens of ms here, so the pass runs in the background with a heartbeat (gpurun kills a silent command after 7 min).
if [ -z "$SKIP_PMC32" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout -k 10 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc32_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 32 --mel-tokens 24 --text-tokens 332 --no-graph --no-cpu-baseline --no-also > $out/pmc32_$c.json 2> $out/pmc32_$c.err) &
  pid=$!
  while kill -0 $pid 2>/dev/null; do echo "[pmc32 $c] running $(date +%T)"; sleep 45; done
  f=$(find $out/pmc32_$c -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" $c $out/pmc32_${c}_step.json > $out/pmc32_$c.txt
  rm -rf $out/pmc32_$c
  head -12 $out/pmc32_$c.txt | cut -c1-170
done
python3 - <<PY
import json
f = json.load(open("$out/pmc32_FETCH_SIZE_step.json")); w = json.load(open("$out/pmc32_WRITE_SIZE_step.json"))
D, NL, V, B, L, T = 1280, 24, 8194, 64, 332, 24
alg = (NL * (12 * D * D + 13 * D) + 4 * D + D * V + V) * 2 + B * 2 * NL * D * 2 * ((32 + L + 2 + 1) + T / 2.0)
hbm = (2.0 * f["per_step_units"] + w["per_step_units"]) * 1024
import subprocess
head = subprocess.run(["git", "-C", "$GRAFT_REPO_ROOT", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "$(cat $GRAFT_REPO_ROOT/.git_head 2>/dev/null)"
json.dump({"kernels_head": head, "engine": "0", "mel_tokens": T, "text_tokens": L, "mean_S": (32 + L + 2 + 1) + T / 2.0, "decode_rows": B, "fetch_kib_raw_per_step": f["per_step_units"], "write_kib_per_step": w["per_step_units"],
           "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": hbm / alg},
          open("$out/pmc_decode_b32.json", "w"), indent=1)
print(open("$out/pmc_decode_b32.json").read())
PY
fi
if [ -n "$SEGV_PROBE" ]; then
  # the r02 SIGSEGV: ONE full-length pass of the 122-launches-per-step path (58 k counted dispatches) with Python's
  # faulthandler on, so the dump says where the process was - inside a library call of ours, or in the tool's finalisation
  ITTS_ENGINE=0 timeout -k 10 700 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/segv_probe -- python3 -X faulthandler $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-also > $out/segv_probe.json 2> $out/segv_probe.err
  echo "segv probe exit code $?" | tee -a $out/segv_probe.err
  ls -la $out/segv_probe/*/ 2>/dev/null | tail -5 >> $out/segv_probe.err
  rm -rf $out/segv_probe
  tail -30 $out/segv_probe.err | cut -c1-220
fi
[ -n "$ONLY_PMC" ] && exit 0
# BASELINE config 3 (32 utterances per GPU = 64 decode rows): bench line + kernel summary
python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline > $out/bench_b32.json 2> $out/bench_b32.err
tail -1 $out/bench_b32.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats32 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline > $out/bench_b32_profiled.json 2> $out/prof32.err
t=$(find $out/stats32 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary_b32.txt
cp $(find $out/stats32 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_b32.csv
rm -rf $out/stats32
head -8 $out/kernel_summary_b32.txt | cut -c1-200
