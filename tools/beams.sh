# the reference's default generate() mode: beam-sample, 3 beams per sentence
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also "$@" 2>gpurun_beams.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'], 'rows', j['config']['decode_batch'])"; }
run --sentences 1 --beams 3
run --sentences 2 --beams 3
run --sentences 4 --beams 3
run --sentences 1
run --sentences 4
