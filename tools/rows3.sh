cd /tmp
for ns in 3; do timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 1 --sentences $ns --mel-tokens 240 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', $ns, 'audio-s/s', j['value'], 'ms/decode step', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])"; done
