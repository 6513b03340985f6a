#!/usr/bin/env python3
"""ONE command that re-collects the hardware counters behind every `roofline.traffic` of bench.py, on the library that is in
the tree right now, and stamps each entry of profiles/pmc_traffic.json with that library's source digest.

    python profiles/collect_pmc.py [--round r06] [--legs headline:f64,headline:f32,cfg3:f64,cfg5:f64,cfg4:f64]

(on the GPU box: `gpurun -- 'python profiles/collect_pmc.py'`).  Per leg it runs profiles/pmc_eval.py (20 loss+gradient
evaluations of that leg's workload) under rocprofv3 in SEPARATE --pmc passes, as MI355X_MICROARCH.md prescribes (FETCH_SIZE
and WRITE_SIZE do not fit one pass; no tracing domain beside --kernel-trace):
    pass 1  FETCH_SIZE
    pass 2  WRITE_SIZE
    pass 3  SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA   (matrix pipe)
    pass 4  SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY                     (issue)
and writes
    gpurun_out/<round>_pmc_<leg>_<dtype>.txt     per-kernel averages of every counter (copy to profiles/)
    profiles/pmc_traffic.json                    one entry per leg: traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch (the
                                                 guide's gfx950 correction), the SQ counters beside it, `sources_sha256` =
                                                 digest of the kernel sources the profiled library was built from
bench.py compares that digest with the library it runs (`traffic_provenance.library_matches`)."""
import argparse
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pinns-tf2.0_amd"))
PASSES = [("FETCH_SIZE",), ("WRITE_SIZE",),
          ("SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA"),
          ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")]
SKIP = ("k_reduce", "k_adam", "k_lbc", "k_lbfgs", "k_zero", "k_pack", "k_err", "k_pick")
POINTS = {"headline": 10000, "cfg3": 10000, "cfg5": 1000000, "cfg4": 20000}


def per_kernel(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    return {(n.split("(")[0].replace("void ", ""), c): (cnt, avg) for n, c, cnt, avg in rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r06")
    ap.add_argument("--legs", default="headline:f64,headline:f32,cfg3:f64,cfg5:f64,cfg4:f64")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    import pinn_native
    pinn_native.load()
    digest = pinn_native.library_digest() or pinn_native._source_digest()
    stale = pinn_native._stale()
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    table_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(table_path))
    for item in args.legs.split(","):
        leg, dtype = item.split(":")
        merged = {}
        for i, ctrs in enumerate(PASSES):
            d = os.path.join("/tmp", "pinn_pmc_%s_%s_%d" % (leg, dtype, i))
            shutil.rmtree(d, ignore_errors=True)
            cmd = ["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "-d", d, "-o", "p", "--", sys.executable,
                                                         os.path.join(ROOT, "profiles", "pmc_eval.py"), leg, dtype]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            if res.returncode != 0 or not dbs:
                print("pass %s of %s:%s failed: %s" % (ctrs, leg, dtype, (res.stdout + res.stderr)[-400:]), flush=True)
                continue
            merged.update(per_kernel(dbs[0]))
            shutil.rmtree(d, ignore_errors=True)
        # the kernels of an evaluation run 20 times each; set-up kernels (copies, fills, the weight cast) run once or a few times
        kernels = sorted({k for (k, c), (cnt, _) in merged.items()
                          if c == "FETCH_SIZE" and cnt >= 20 and not any(s in k for s in SKIP) and not k.startswith("__amd_rocclr")})
        if not kernels:
            print("no counters for %s:%s" % (leg, dtype), flush=True)
            continue
        lines = ["# %s %s: per-kernel averages over the launches of 20 loss+gradient evaluations (profiles/pmc_eval.py), "
                 "separate rocprofv3 --pmc passes; library sources-sha256 %s" % (leg, dtype, digest),
                 "%-52s %-30s %6s %16s" % ("kernel", "counter", "calls", "avg")]
        for (k, c), (cnt, avg) in sorted(merged.items()):
            if cnt < 20 or k.startswith("__amd_rocclr"):
                continue
            lines.append("%-52s %-30s %6d %16.2f" % (k[:52], c, cnt, avg))
        open(os.path.join(args.out, "%s_pmc_%s_%s.txt" % (args.round, leg, dtype)), "w").write("\n".join(lines) + "\n")
        get = lambda c: sum(merged.get((k, c), (0, 0.0))[1] for k in kernels)
        fetch, write = get("FETCH_SIZE"), get("WRITE_SIZE")
        eng_path = {("headline", "f64"): 7, ("headline", "f32"): 2, ("cfg3", "f64"): 7, ("cfg3", "f32"): 2, ("cfg5", "f64"): 7,
                    ("cfg5", "f32"): 2, ("cfg4", "f64"): 8, ("cfg4", "f32"): 3}[(leg, dtype)]
        entry = {"leg": leg, "dtype": dtype, "kernel_path": eng_path, "points": POINTS[leg],
                 "traffic_bytes_per_launch": round((2.0 * fetch + write) * 1024.0), "kernel": " + ".join(kernels),
                 "fetch_size_kb_raw": round(fetch, 2), "write_size_kb": round(write, 2),
                 "sq": {c: get(c) for p in PASSES[2:] for c in p},
                 "source": "profiles/%s_pmc_%s_%s.txt" % (args.round, leg, dtype), "sources_sha256": digest,
                 "library_was_stale": bool(stale)}
        table["entries"] = [e for e in table["entries"] if (e["leg"], e["dtype"], e["kernel_path"], e["points"]) !=
                            (leg, dtype, eng_path, POINTS[leg])] + [entry]
        json.dump(table, open(table_path, "w"), indent=1)
        shutil.copy(table_path, os.path.join(args.out, "%s_pmc_traffic.json" % args.round))
        print(json.dumps(entry), flush=True)


if __name__ == "__main__":
    main()
