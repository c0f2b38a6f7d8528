#!/bin/bash
# kernel summary of the 64-row decode step (BASELINE config 3), short run:  tools/prof_b32.sh tag [extra bench flags]
tag=${1:-b32}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 32 --mel-tokens 160 --no-cpu-baseline --no-also "$@" > $out/bench.json 2> $out/prof.err
t=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary.txt
rm -rf $out/stats
head -14 $out/kernel_summary.txt | cut -c1-230
