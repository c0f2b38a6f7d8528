# quick A/B of the two headline configs: audio-s/s and ms per decode step
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'], 'rows', j['config']['decode_batch'])"; }
run
run --batch 32
run --batch 4
