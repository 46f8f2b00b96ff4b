#!/bin/bash
# rocprofv3 passes of the round (run on the GPU box through gpurun): kernel trace + SQ / LDS / FETCH_SIZE /
# WRITE_SIZE counters of bench.py's decoder launch and of the evaluation kernels (tools/prof_eval_kernels.py).
# Text summaries land in gpurun_out/prof/<tag>_*.txt (tools/rocpd_summary.py); copy the ones to keep into profiles/.
set -u
TAG=${1:-r02}
PASSES=${2:-all}          # all | decoder | eval
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
EVAL="python3 $ROOT/tools/prof_eval_kernels.py"
run() {  # name, command..., then counters
    local name=$1; shift
    local cmd=$1; shift
    rm -rf /tmp/prof_$name
    if [ $# -eq 0 ]; then
        timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name/trace -- $cmd > /tmp/prof_$name.log 2>&1
    else
        timeout 300 rocprofv3 --pmc "$@" -d /tmp/prof_$name/pmc -- $cmd > /tmp/prof_$name.log 2>&1
    fi
    python3 $ROOT/tools/rocpd_summary.py /tmp/prof_$name > $OUT/${TAG}_$name.txt 2>&1
    rm -rf /tmp/prof_$name
}
if [ "$PASSES" != eval ]; then
run decoder_trace "$BENCH"
run decoder_sq "$BENCH" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run decoder_lds "$BENCH" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE
run decoder_fetch "$BENCH" FETCH_SIZE
run decoder_write "$BENCH" WRITE_SIZE
fi
if [ "$PASSES" != decoder ]; then
run eval_trace "$EVAL"
run eval_sq "$EVAL" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run eval_fetch "$EVAL" FETCH_SIZE
run eval_write "$EVAL" WRITE_SIZE
fi
ls -la $OUT
