#!/bin/bash
# Every kernel between the decoder prologue and the end of the grid launch of the last vox-64 inference of tools/inference_once.py
# (start / end relative to the prologue's first kernel, HW queue): where the time between the encoder and the grid launch goes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptl
rocprofv3 --kernel-trace -d /tmp/ptl/trace -- python3 $ROOT/tools/inference_once.py 1 ${1:-64} > $OUT/inference_timeline_run.txt 2>&1
python3 - > $OUT/inference_timeline.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/ptl/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, grid_x, queue_id from kernels order by start"))
last = max(i for i, r in enumerate(rows) if "sdf_decode_split_kernel<true>" in r[0])
first = max(i for i, r in enumerate(rows[:last]) if "lat_linear_kernel<false, false, false, true>" in r[0])
t0 = rows[first][1]
for n, s, e, g, q in rows[first - 3:last + 4]:
    print("%-70s grid %7d q %2s  %9.1f .. %9.1f us (%.1f)" % (n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70], g, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
tail -1 $OUT/inference_timeline_run.txt
cat $OUT/inference_timeline.txt
