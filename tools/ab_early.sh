# c_fc requested N x 64 clocks behind c_attn on the workgroups without attention (ITTS_ENGINE_EARLY_FC):  tools/ab_early.sh
cd /tmp
run() { timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EARLY_FC=${ITTS_ENGINE_EARLY_FC:-0} $*', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'])"; }
for e in 55 62 70 78 86; do ITTS_ENGINE_EARLY_FC=$e run; done
for r in 1 3 4 6; do for e in 0 55 70 85; do ITTS_ENGINE_EARLY_FC=$e run --sentences $r; done; done
for e in 0 70; do ITTS_ENGINE_EARLY_FC=$e run --sentences 1 --beams 3; done
