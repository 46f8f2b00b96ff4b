#!/bin/bash
# kernel trace of the training step (tools/train_cpu_gpu.py), summarised per kernel and per (kernel, grid)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=${1:-train}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_t
rocprofv3 --kernel-trace --stats -d /tmp/pt_t/trace -- python3 $ROOT/tools/train_cpu_gpu.py > $OUT/${TAG}_run.txt 2>&1
python3 $ROOT/tools/rocpd_summary.py /tmp/pt_t > $OUT/${TAG}_summary.txt 2>&1
python3 - > $OUT/${TAG}_by_grid.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/pt_t/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("# kernels view columns:", cols)
rows = list(cur.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels "
                        "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 150"))
for r in rows:
    print("%-60s grid %7d %5d %5d  calls %5d  total_us %10.1f  avg_us %8.2f" % (r[0][:60], r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3))
PY
tail -2 $OUT/${TAG}_run.txt; head -70 $OUT/${TAG}_by_grid.txt
