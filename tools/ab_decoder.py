#!/usr/bin/env python3
"""Same-box A/B of decoder libraries: mean launch time of the 129^3 grid per library, alternating.
    python tools/ab_decoder.py libA.so libB.so [...]      (paths relative to the repository root)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ZS_LIB_PATH=os.path.join(ROOT, lib))
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline",
                                       "--steps", "10"], env=env, stderr=subprocess.DEVNULL)
        d = json.loads(out.decode().strip().split("\n")[-1])
        res.setdefault(lib, []).append(d["roofline"]["launch_ms_mean"])
for lib, v in res.items():
    print("%-40s launch ms: %s" % (lib, " ".join("%.3f" % x for x in v)))
