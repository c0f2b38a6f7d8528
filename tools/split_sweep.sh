# batched decode (64 rows): K split of the residual projections (absorbed by the LayerNorm that follows)
cd /tmp
run() { timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['roofline']['avg_launch_ms'])"; }
ITTS_SKINNY_SPLIT_PROJ=2 ITTS_SKINNY_SPLIT_PROJ2=4 run "proj 2, proj2 4"
run "proj 3, proj2 5 (the code default)"
ITTS_SKINNY_SPLIT_PROJ=6 ITTS_SKINNY_SPLIT_PROJ2=6 run "proj 6, proj2 6"
ITTS_SKINNY_SPLIT_PROJ=4 ITTS_SKINNY_SPLIT_PROJ2=8 run "proj 4, proj2 8"
ITTS_SKINNY_SPLIT_PROJ=6 ITTS_SKINNY_SPLIT_PROJ2=4 run "proj 6, proj2 4"
ITTS_SKINNY_SPLIT_PROJ=3 ITTS_SKINNY_SPLIT_PROJ2=5 run "proj 3, proj2 5"
