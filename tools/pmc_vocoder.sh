#!/bin/bash
# is Activation1d (snake_aa_lds_kernel) / conv_lds_kernel VALU-issue bound?  SQ counters of the vocoder kernels
# of the default workload (eager launches; one pass per counter group: 8 SQ slots, GRBM apart).
#   tools/pmc_vocoder.sh r03
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pv_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-also > $out/pv_$name.json 2> $out/pv_$name.err
  f=$(find $out/pv_$name -name "*counter_collection.csv" | head -1)
  t=$(find $out/pv_$name -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_vocoder.py "$f" "$t" > $out/pmc_vocoder_$name.txt
  rm -rf $out/pv_$name
  cat $out/pmc_vocoder_$name.txt | cut -c1-230
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA
run grbm GRBM_GUI_ACTIVE
