# decode step of the small-batch configurations (ms):  tools/ab_rows.sh
cd /tmp
run() { timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'], 'rows', j['config']['decode_batch'])"; }
run --sentences 1
run
run --sentences 3
run --sentences 4
run --sentences 1 --beams 3
run --sentences 2 --beams 3
