#!/bin/bash
# Round-6 rocprofv3 passes (one gpurun call): the decoder launch of one rank of an 8-GPU step (batch 8: eight programs in
# one launch) next to the single-GPU launch - kernel trace + FETCH_SIZE + WRITE_SIZE (separate --pmc passes).
# Summaries: gpurun_out/prof/<tag>_*.txt (tools/rocpd_summary.py); copy the ones to keep into profiles/.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, command, then counters
    local name=$1; shift
    local cmd=$1; shift
    rm -rf /tmp/prof_$name
    if [ $# -eq 0 ]; then
        timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name/trace -- $cmd > /tmp/prof_$name.log 2>&1
    else
        timeout 300 rocprofv3 --pmc "$@" -d /tmp/prof_$name/pmc -- $cmd > /tmp/prof_$name.log 2>&1
    fi
    python3 $ROOT/tools/rocpd_summary.py /tmp/prof_$name > $OUT/${TAG}_$name.txt 2>&1
    tail -2 /tmp/prof_$name.log >> $OUT/${TAG}_$name.txt
    rm -rf /tmp/prof_$name
}
for W in 8 1; do
    V="python3 $ROOT/tools/prof_vranks.py $W 0 5"
    run vranks${W}_trace "$V"
    run vranks${W}_fetch "$V" FETCH_SIZE
    run vranks${W}_write "$V" WRITE_SIZE
done
ls -la $OUT
