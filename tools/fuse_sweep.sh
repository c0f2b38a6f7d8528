# A/B of the fused projection+attention launch (decode batches <= 4): audio-s/s and ms per decode step for the two-launch
# path and for the fused launch at several poll timings.  Run on the GPU box: bash tools/fuse_sweep.sh
cd /tmp
run() { timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['roofline']['avg_launch_ms'])"; }
ITTS_FUSE_QKV_ATTN=0 run unfused
ITTS_FUSE_QKV_ATTN=1 ITTS_FUSE_SLEEP0=24 ITTS_FUSE_SLEEP1=4 run s0=24,s1=4
ITTS_FUSE_QKV_ATTN=1 ITTS_FUSE_SLEEP0=32 ITTS_FUSE_SLEEP1=4 run s0=32,s1=4
ITTS_FUSE_QKV_ATTN=1 ITTS_FUSE_SLEEP0=40 ITTS_FUSE_SLEEP1=4 run s0=40,s1=4
ITTS_FUSE_QKV_ATTN=1 ITTS_FUSE_SLEEP0=48 ITTS_FUSE_SLEEP1=2 run s0=48,s1=2
ITTS_FUSE_QKV_ATTN=1 ITTS_FUSE_SLEEP0=40 ITTS_FUSE_SLEEP1=16 run s0=40,s1=16
ITTS_FUSE_QKV_ATTN=0 run unfused_again
