#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter (rocpd SQLite output).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o f -- python bench.py ...   (one counter group per pass)
    python profiles/summarize_pmc.py DIR/f_results.db [DIR2/w_results.db ...]
FETCH_SIZE / WRITE_SIZE are kilobytes at the L2 <-> fabric interface.  MI355X_MICROARCH.md: on gfx950
FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads by exactly 2x; other widths and
WRITE_SIZE are uncalibrated -- both the raw and the x2 figure are printed for FETCH_SIZE."""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                      "from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    print("# %s" % path)
    print("%-44s %-11s %6s %12s %12s %12s" % ("kernel", "counter", "calls", "avg_KB", "min_KB", "max_KB"))
    for name, ctr, n, avg, lo, hi in rows:
        short = name.split("(")[0].replace("void ", "")
        extra = "   (x2 = %.1f KB if wide-streaming)" % (2 * avg) if ctr == "FETCH_SIZE" else ""
        print("%-44s %-11s %6d %12.2f %12.2f %12.2f%s" % (short[:44], ctr, n, avg, lo, hi, extra))
    print()
