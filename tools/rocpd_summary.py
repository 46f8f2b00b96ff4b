#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into a small text file for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r1 > profiles/r01_decoder_summary.txt

Expects sub-directories trace/ (--kernel-trace --stats) and pmc_*/ (--pmc passes), each
holding <dir>/<host>/<pid>_results.db as written by `rocprofv3 -d <dir>`.
"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute(sql))


def main(root):
    for sub in sorted(os.listdir(root)):
        dbs = glob.glob(os.path.join(root, sub, "*", "*_results.db"))
        if not dbs:
            continue
        db = dbs[0]
        print("== %s (%s)" % (sub, os.path.relpath(db, root)))
        tabs = [r[0] for r in q(db, "select name from sqlite_master where type in ('table','view')")]
        if "top_kernels" in tabs:
            print("kernel-trace --stats  (durations in ns)")
            print("%-72s %6s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
            for name, calls, tot, avg, pct in q(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
                print("%-72s %6d %14.0f %12.0f %7.3f" % (name[:72], calls, tot * 1e3 if tot < 1e8 else tot,
                                                        avg * 1e3 if tot < 1e8 else avg, pct))
        rows = q(db, "select kernel_name,counter_name,count(*),avg(value),min(value),max(value),"
                     "max(vgpr_count),max(accum_vgpr_count),max(sgpr_count),max(scratch_size),max(lds_block_size) "
                     "from counters_collection group by kernel_name,counter_name") if "counters_collection" in tabs else []
        if rows:
            print("PMC per dispatch: kernel, counter, dispatches, avg, min, max | vgpr agpr sgpr scratch lds")
            for r in rows:
                print("%-60s %-26s %4d %16.1f %16.1f %16.1f | %s" % (r[0][:60], r[1], r[2], r[3], r[4], r[5],
                                                                  " ".join(str(x) for x in r[6:])))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
