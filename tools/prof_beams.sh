#!/bin/bash
# kernel trace of the 3-beam (reference default generate() mode) bench: tools/prof_beams.sh <tag>
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_beams -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --beams 3 --no-cpu-baseline --no-also > $out/bench_beams_profiled.json 2> $out/prof_beams.err
t=$(find $out/stats_beams -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary_beams.txt
rm -rf $out/stats_beams
head -16 $out/kernel_summary_beams.txt | cut -c1-190
tail -1 $out/bench_beams_profiled.json | cut -c1-400
