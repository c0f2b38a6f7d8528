# batched decode attention: register window (ATTN_NIT_MANY = 4 / 6 / 8 key pairs) - measured: no difference (1.536 / 1.550 / 1.536 ms).
# The variants are whole libraries built beforehand (they travel with the snapshot, then delete them):
#   cd index-tts-ipex_amd/csrc && mkdir -p var && for n in 4 6; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
#     -DATTN_NIT_MANY=$n -c decode2.hip -o /tmp/d2_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o var/libitts_n$n.so \
#     $(ls *.o | grep -v decode2.o) /tmp/d2_$n.o; done
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['roofline']['avg_launch_ms'])"; }
run "NIT 8"
for n in 4 6; do ITTS_HIP_LIB=$GRAFT_REPO_ROOT/index-tts-ipex_amd/csrc/var/libitts_n$n.so run "NIT $n"; done
