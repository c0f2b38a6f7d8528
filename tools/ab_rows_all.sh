# decode step of every small-batch configuration, persistent engine (E=1) against the launch path (E=0):  tools/ab_rows_all.sh
cd /tmp
run() { timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=${ITTS_ENGINE:-1} $*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'], 'rows', j['config']['decode_batch'])"; }
for r in 1 2 3 4 5 6; do run --sentences $r; ITTS_ENGINE=0 run --sentences $r; done
run --sentences 1 --beams 3; ITTS_ENGINE=0 run --sentences 1 --beams 3
run --sentences 2 --beams 3; ITTS_ENGINE=0 run --sentences 2 --beams 3
run --sentences 1 --sample; ITTS_ENGINE=0 run --sentences 1 --sample
