#!/bin/bash
# usage: prof.sh <outdir-name> [bench args...]
name=$1; shift
mkdir -p gpurun_out/$name && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name/bench.log 2>&1
cd $GRAFT_REPO_ROOT
t=$(find gpurun_out/$name -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py "$t" > gpurun_out/$name/summary.txt
find gpurun_out/$name -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/$name/bench.log | cut -c1-400
cat gpurun_out/$name/summary.txt
