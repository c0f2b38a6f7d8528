# gather pacing of the persistent engine (s_sleep units of 64 clocks) at a given row count:  tools/eng_pacing_rows.sh [rows]
cd /tmp
R=${1:-1}
run() { timeout -k 10 120 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --sentences $R 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows $R FD=${ITTS_ENGINE_FIRST_DELAY:-d} AD=${ITTS_ENGINE_ACT_DELAY:-d} EF=${ITTS_ENGINE_EARLY_FC:-d}', '->', 'ms/decode step', j['roofline']['avg_launch_ms'])"; }
run
for fd in 8 11 18 22; do ITTS_ENGINE_FIRST_DELAY=$fd run; done
for ad in 6 10 22 28; do ITTS_ENGINE_ACT_DELAY=$ad run; done
for ef in 40 74 100; do ITTS_ENGINE_EARLY_FC=$ef run; done
run
