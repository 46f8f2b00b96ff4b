#!/bin/bash
# PMC passes of the 256 x 256 ping-pong GEMM kernel (and the 128 x 128 kernel it replaces: ZS_CONV_PP256=0) on one
# pointwise shape: cache behaviour (TCC hit / miss, memory-side requests), SQ occupancy, LDS.
#   tools/prof_pp256.sh <tag> <M> <K> <N>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=${1:-pp256}; M=${2:-5516}; K=${3:-768}; N=${4:-3072}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_conv.py --B 1 --H $M --W 1 --Cin $K --Cout $N --k 1 --mode fwd --iters 10 --engine ops"
run() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pp_$name
  rocprofv3 --pmc "$@" -d /tmp/pp_$name/pmc -- $CMD > /dev/null 2>&1
  python3 $ROOT/tools/rocpd_summary.py /tmp/pp_$name 2>&1 | grep "pp256\|conv_gemm_dma" >> $OUT/${TAG}.txt
}
echo "# $TAG: M=$M K=$K N=$N  ZS_CONV_PP256=${ZS_CONV_PP256:-1}" > $OUT/${TAG}.txt
rm -rf /tmp/pp_t
rocprofv3 --kernel-trace --stats -d /tmp/pp_t/trace -- $CMD 2>&1 | grep TFLOP >> $OUT/${TAG}.txt
python3 $ROOT/tools/rocpd_summary.py /tmp/pp_t 2>&1 | head -6 >> $OUT/${TAG}.txt
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
cat $OUT/${TAG}.txt
