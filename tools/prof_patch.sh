#!/bin/bash
# PMC passes of the input-patch 3x3 kernel on the DPT head layer (128 -> 32 at 224 x 224 x 28): where do the cycles of the
# vector-memory path go (TA / TCP / TCC / UTCL1), next to the SQ view.  Summaries -> gpurun_out/prof/<tag>_*.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=${1:-patch}
SHAPE=${2:-128,32,224}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_patch.py --iters 3 --shape $SHAPE"
pass() {
  local name=$1; shift
  rm -rf /tmp/pp_$name
  timeout 120 rocprofv3 --pmc "$@" -d /tmp/pp_$name/pmc -- $CMD > /dev/null 2>&1
  python3 $ROOT/tools/rocpd_summary.py /tmp/pp_$name 2>&1 | grep -E "patch|conv_gemm_dma" | head -12 > $OUT/${TAG}_$name.txt
  rm -rf /tmp/pp_$name
}
rm -rf /tmp/pp_t
rocprofv3 --kernel-trace --stats -d /tmp/pp_t/trace -- $CMD > $OUT/${TAG}_run.txt 2>&1
python3 $ROOT/tools/rocpd_summary.py /tmp/pp_t 2>&1 | head -8 > $OUT/${TAG}_trace.txt
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS
pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_CYCLE_sum
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
cat $OUT/${TAG}_run.txt | grep Cout; cat $OUT/${TAG}_trace.txt $OUT/${TAG}_sq.txt $OUT/${TAG}_sq2.txt $OUT/${TAG}_ta.txt $OUT/${TAG}_tcp.txt $OUT/${TAG}_tcc.txt $OUT/${TAG}_tlb.txt
