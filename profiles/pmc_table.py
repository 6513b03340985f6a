#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_eval.py -> one entry of profiles/pmc_traffic.json
(the table bench.py reads `roofline.traffic` from).

    python profiles/pmc_table.py <leg> <dtype> <kernel_path> <points> F_results.db W_results.db [source note]

traffic per launch = sum over the kernels of ONE loss+gradient evaluation (the reduction / optimiser kernels excluded) of
2 x FETCH_SIZE + WRITE_SIZE, per-kernel averages over the run's calls: MI355X_MICROARCH.md -- on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads; WRITE_SIZE is taken as it is; both in KB at the L2 <-> fabric interface."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("k_reduce", "k_adam", "k_lbc", "k_lbfgs", "k_zero", "k_pack", "k_err", "k_pick")


def per_kernel(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    return {(n.split("(")[0].replace("void ", ""), c): (cnt, avg) for n, c, cnt, avg in rows}


def main():
    leg, dtype, path, points, fdb, wdb = sys.argv[1:7]
    note = sys.argv[7] if len(sys.argv) > 7 else "%s, %s" % (os.path.basename(fdb), os.path.basename(wdb))
    f, w = per_kernel(fdb), per_kernel(wdb)
    kernels = sorted({k for k, c in f if c == "FETCH_SIZE" and not any(s in k for s in SKIP)})
    fetch = sum(f[(k, "FETCH_SIZE")][1] for k in kernels)
    write = sum(w.get((k, "WRITE_SIZE"), (0, 0.0))[1] for k in kernels)
    entry = {"leg": leg, "dtype": dtype, "kernel_path": int(path), "points": int(points),
             "traffic_bytes_per_launch": round((2.0 * fetch + write) * 1024.0), "kernel": " + ".join(kernels),
             "fetch_size_kb_raw": round(fetch, 2), "write_size_kb": round(write, 2), "source": note}
    table = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    t = json.load(open(table))
    t["entries"] = [e for e in t["entries"] if (e["leg"], e["dtype"], e["kernel_path"], e["points"]) !=
                    (leg, dtype, int(path), int(points))] + [entry]
    json.dump(t, open(table, "w"), indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
