#!/bin/bash
# kernel summary of the batched vocoder (64 sentences x 480 frames = BASELINE config 3):  tools/prof_vocoder.sh tag
tag=${1:-voc}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/tools/vocoder_only.py 64 480 2 > $out/run.txt 2> $out/prof.err
t=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$t" > $out/kernel_summary.txt
rm -rf $out/stats
cat $out/run.txt; head -40 $out/kernel_summary.txt | cut -c1-200
