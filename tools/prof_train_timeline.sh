#!/bin/bash
# Kernel trace of the SEGMENTED captured training step (tools/train_segments.py segments): where, inside the last step's
# backward window, the bucket launches fall (copy_multi_kernel = a bucket's pack, issued together with its all-reduce; RCCL
# kernels appear when the group has more than one rank - a one-rank in-place all_reduce launches nothing).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptt
rocprofv3 --kernel-trace -d /tmp/ptt/trace -- python3 $ROOT/tools/train_segments.py segments 4 > $OUT/train_timeline_run.txt 2>&1
python3 - > $OUT/train_timeline.txt 2>&1 <<PY
import glob, sqlite3
db = glob.glob("/tmp/ptt/trace/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r[0]]
lo, hi = adam[-2], adam[-1]                      # the last whole step: behind the previous AdamW up to this one
step = rows[lo + 1:hi + 1]
t0 = step[0][1]
print("last step: %d kernels, %.2f ms from its first kernel to the end of AdamW" % (len(step), (step[-1][2] - t0) / 1e6))
marks = [(n, s, e) for n, s, e in step if "copy_multi" in n or "nccl" in n.lower() or "rccl" in n.lower() or "adamw" in n
         or "bce" in n.lower() or "shape_loss" in n.lower()]
for n, s, e in marks:
    print("%-60s start %8.3f ms  end %8.3f ms" % (n.replace("(anonymous namespace)::", "")[:60], (s - t0) / 1e6, (e - t0) / 1e6))
PY
cat $OUT/train_timeline_run.txt | tail -3
cat $OUT/train_timeline.txt
