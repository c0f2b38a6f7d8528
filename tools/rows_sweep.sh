# decode step time against the number of rows in the decode batch (2 rows per utterance)
cd /tmp
for b in 1 2 3 4 6 8 16; do timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch $b --mel-tokens 240 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', 2*$b, 'audio-s/s', j['value'], 'ms/decode step', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])"; done
