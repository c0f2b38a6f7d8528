#!/bin/bash
# Round profile: (1) bench with cpu baseline, (2) rocprofv3 --kernel-trace --stats of the same bench command,
# (3) PMC passes (FETCH_SIZE, WRITE_SIZE) on a short decode for the HBM-traffic column.  Run on the GPU box:
#   tools/round_profile.sh r01
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/prof.err
t=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats
head -12 $out/kernel_summary.txt | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --mel-tokens 64 --no-cpu-baseline > $out/pmc_$c.json 2> $out/pmc_$c.err
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" $c > $out/pmc_$c.txt
  rm -rf $out/pmc_$c
  cat $out/pmc_$c.txt | head -12
done
