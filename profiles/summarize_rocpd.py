#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the plain-text per-kernel summary committed under profiles/.

    python profiles/summarize_rocpd.py gpurun_out/prof/NAME_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# durations in microseconds")
    print("%-64s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0].replace("void ", "")
        print("%-64s %8d %14.1f %12.3f %6.2f%%" % (short[:64], calls, total, avg, pct))
    print()
    print("# per-kernel launch geometry / registers (first dispatch of each kernel)")
    print("%-64s %10s %6s %6s %6s %8s %8s" % ("kernel", "grid", "wg", "vgpr", "sgpr", "lds", "scratch"))
    seen = set()
    for row in cur.execute("select name,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,"
                           "scratch_size from kernels order by start"):
        if row[0] in seen:
            continue
        seen.add(row[0])
        short = row[0].split("(")[0].replace("void ", "")
        print("%-64s %10d %6d %6d %6d %8d %8d" % ((short[:64],) + tuple(row[1:])))


if __name__ == "__main__":
    main(sys.argv[1])
