# gather pacing of the persistent engine (s_sleep units of 64 clocks):  tools/eng_pacing_rows.sh
cd /tmp
run() { timeout -k 10 120 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FD=${ITTS_ENGINE_FIRST_DELAY:-22} AD=${ITTS_ENGINE_ACT_DELAY:-20} CD=${ITTS_ENGINE_CTX_DELAY:-0} PS=${ITTS_ENGINE_PASS_SLEEP:-1} $*', '->', 'ms/decode step', j['roofline']['avg_launch_ms'])"; }
for fd in 10 13 16 19 22; do for ad in 4 10 16 20; do ITTS_ENGINE_FIRST_DELAY=$fd ITTS_ENGINE_ACT_DELAY=$ad run --sentences 2; done; done
ITTS_ENGINE_FIRST_DELAY=16 ITTS_ENGINE_ACT_DELAY=10 ITTS_ENGINE_PASS_SLEEP=0 run --sentences 2
ITTS_ENGINE_FIRST_DELAY=16 ITTS_ENGINE_ACT_DELAY=10 ITTS_ENGINE_PASS_SLEEP=2 run --sentences 2
ITTS_ENGINE_FIRST_DELAY=16 ITTS_ENGINE_ACT_DELAY=10 run --sentences 1
ITTS_ENGINE_FIRST_DELAY=22 ITTS_ENGINE_ACT_DELAY=20 run --sentences 1
