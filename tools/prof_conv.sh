#!/bin/bash
# PMC + trace of the forward convolution engine on two large layer shapes (ViT fc1 at batch 28; 3x3 256->256 at 56x56 x 28)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=${1:-conv}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ZS_CONV_STREAM_K=${ZS_CONV_STREAM_K:-0}
for shape in "pw --B 28 --H 14 --Cin 768 --Cout 3072 --k 1" "c3 --B 28 --H 56 --Cin 256 --Cout 256 --k 3"; do
  set -- $shape; name=$1; shift
  CMD="python3 $ROOT/tools/bench_conv.py $* --mode fwd --iters 20 --engine ops"
  rm -rf /tmp/pc_*
  rocprofv3 --kernel-trace --stats -d /tmp/pc_t/trace -- $CMD > $OUT/${TAG}_${name}_run.txt 2>&1
  python3 $ROOT/tools/rocpd_summary.py /tmp/pc_t 2>&1 | head -8 > $OUT/${TAG}_${name}_trace.txt
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pc_a/pmc -- $CMD > /dev/null 2>&1
  python3 $ROOT/tools/rocpd_summary.py /tmp/pc_a 2>&1 | grep "conv_gemm\|conv3x3_patch" | head -30 > $OUT/${TAG}_${name}_sq.txt
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS -d /tmp/pc_b/pmc -- $CMD > /dev/null 2>&1
  python3 $ROOT/tools/rocpd_summary.py /tmp/pc_b 2>&1 | grep "conv_gemm\|conv3x3_patch" | head -30 > $OUT/${TAG}_${name}_lds.txt
done
cat $OUT/${TAG}_*_run.txt | grep TFLOP; cat $OUT/${TAG}_*_trace.txt $OUT/${TAG}_*_sq.txt $OUT/${TAG}_*_lds.txt
