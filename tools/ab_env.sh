# A/B of environment settings in one gpurun call: tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (two rounds, interleaved)
cd /tmp
args="$1"; shift
for round in 1 2; do
for kv in "$@"; do
  env $kv timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also $args 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', '$args', '->', j['value'], 'audio-s/s, ms/decode step', j['roofline']['avg_launch_ms'], 'rows', j['config']['decode_batch'], j.get('decode_mode'))"
done
done
