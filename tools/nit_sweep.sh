# batched decode attention: register window (ATTN_NIT_MANY) variants built into csrc/var/libitts_n<N>.so
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['roofline']['avg_launch_ms'])"; }
run "NIT 8"
for n in 4 6; do ITTS_HIP_LIB=$GRAFT_REPO_ROOT/index-tts-ipex_amd/csrc/var/libitts_n$n.so run "NIT $n"; done
