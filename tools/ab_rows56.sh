# 5 / 6 decode rows, engine vs launch path, and the engine's gather pacing there:  tools/ab_rows56.sh
cd /tmp
run() { timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=${ITTS_ENGINE:-1} FD=${ITTS_ENGINE_FIRST_DELAY:-d} AD=${ITTS_ENGINE_ACT_DELAY:-d} $*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'])"; }
for r in 5 6; do run --sentences $r; ITTS_ENGINE=0 run --sentences $r; done
run --sentences 2 --beams 3; ITTS_ENGINE=0 run --sentences 2 --beams 3
for fd in 4 8 16; do ITTS_ENGINE_FIRST_DELAY=$fd run --sentences 6; done
for ad in 0 4 16; do ITTS_ENGINE_ACT_DELAY=$ad run --sentences 6; done
