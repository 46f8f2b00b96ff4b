#!/bin/bash
# kernel trace of the encoder half of Graph.forward at one batch size, per kernel and per (kernel, grid)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
B=${1:-1}
TAG=${2:-enc_b$B}
ENC=${3:-resnet}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe_t
rocprofv3 --kernel-trace --stats -d /tmp/pe_t/trace -- python3 $ROOT/tools/prof_encoder.py $B 10 $ENC > $OUT/${TAG}_run.txt 2>&1
python3 $ROOT/tools/rocpd_summary.py /tmp/pe_t > $OUT/${TAG}_summary.txt 2>&1
python3 - > $OUT/${TAG}_by_grid.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/pe_t/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels "
                        "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 140"))
tot = list(cur.execute("select sum(end-start), count(*) from kernels"))[0]
print("# total kernel time %.1f us over %d launches (11 forwards)" % (tot[0] / 1e3, tot[1]))
for r in rows:
    print("%-100s grid %7d %5d %5d  calls %5d  total_us %10.1f  avg_us %8.2f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3))
PY
head -45 $OUT/${TAG}_by_grid.txt
