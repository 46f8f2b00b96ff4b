#!/bin/bash
# kernel trace of any python tool, per (kernel, grid):  bash tools/prof_by_grid.sh TAG tools/some_tool.py [args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
TAG=$1; shift
SCRIPT=$ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbg_t
rocprofv3 --kernel-trace --stats -d /tmp/pbg_t/trace -- python3 $SCRIPT "$@" > $OUT/${TAG}_run.txt 2>&1
python3 - > $OUT/${TAG}_by_grid.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/pbg_t/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                        "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 60"))
tot = list(cur.execute("select sum(end-start), count(*) from kernels"))[0]
print("# total kernel time %.1f us over %d launches" % (tot[0] / 1e3, tot[1]))
for r in rows:
    print("%-70s grid %7d %5d %5d  calls %5d  total_us %10.1f  avg_us %8.2f  min %8.2f  max %8.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3, r[7] / 1e3, r[8] / 1e3))
PY
head -30 $OUT/${TAG}_by_grid.txt
