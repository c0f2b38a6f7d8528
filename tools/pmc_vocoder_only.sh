#!/bin/bash
# Counters of the vocoder kernels at BASELINE config 3's shapes (tools/vocoder_only.py B x 480 frames, eager launches): HBM bytes
# (FETCH_SIZE / WRITE_SIZE, one pass each) and the SQ picture of the fused Activation1d + conv kernels.   tools/pmc_vocoder_only.sh r04 [B]
tag=${1:-rXX}
B=${2:-16}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
sed -i 's/"gemm_mfma"))/"gemm_mfma", "gemm_p8"))/' $GRAFT_REPO_ROOT/tools/pmc_vocoder.py
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pvo_$name -- python3 $GRAFT_REPO_ROOT/tools/vocoder_only.py $B 480 1 > $out/pvo_$name.txt 2> $out/pvo_$name.err
  f=$(find $out/pvo_$name -name "*counter_collection.csv" | head -1)
  t=$(find $out/pvo_$name -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_vocoder.py "$f" "$t" > $out/pmc_vocoder_b${B}_$name.txt
  rm -rf $out/pvo_$name
  head -24 $out/pmc_vocoder_b${B}_$name.txt | cut -c1-230
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA
