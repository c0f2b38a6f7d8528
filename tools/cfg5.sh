# BASELINE config 5 (2000-char long form: 20 sentences as one decode batch, fp8 GPT weights): tiled vs row-major weight stream
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --sentences 20 --gpt-fp8 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"; }
run "config 5, tiled"
ITTS_NO_TILED_W=1 run "config 5, row-major"
