#!/bin/bash
# start / end times of the decoder kernels of the last inference of tools/inference_once.py (does the per-image check overlap the grid launch?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptl
rocprofv3 --kernel-trace -d /tmp/ptl/trace -- python3 $ROOT/tools/inference_once.py "$@" > $OUT/timeline_run.txt 2>&1
python3 - > $OUT/timeline.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/ptl/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, grid_x, queue_id from kernels order by start"))
last = max(i for i, r in enumerate(rows) if "sdf_decode_split_kernel<true>" in r[0])
first = max(i for i, r in enumerate(rows[:last]) if "sdf_decode_kernel<true" in r[0])      # the previous inference's last launch
rows = [r for r in rows[first:last + 2] if (r[2] - r[1]) > 20000 or "sdf_" in r[0] or "lat_" in r[0]] if len(rows) > 400 else rows
t0 = rows[0][1]
prev = t0
for n, s, e, g, q in rows:
    if s - prev > 30000:
        print("      ... gap / short kernels %.1f us" % ((s - prev) / 1e3))
    prev = max(prev, e)
    print("%-44s grid %6d queue %3s  start %9.1f us  end %9.1f us  (%.1f)" % (n.replace("void (anonymous namespace)::", "")[:44], g, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat $OUT/timeline.txt
