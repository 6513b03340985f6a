"""A short run of the dominant kernels for the rocprofv3 --pmc passes (counter collection serialises every dispatch,
so the full bench is far too long under it): 20 loss+gradient evaluations of ONE bench leg's workload per arithmetic,
nothing else.   leg = headline (N_f = 10000) | cfg5 (N_f = 10^6) | cfg3 (identification, N_u = 10000) | cfg4 (Schrodinger)

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o f -- python profiles/pmc_eval.py cfg4 f64
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d DIR -o w -- python profiles/pmc_eval.py cfg4 f64
    python profiles/summarize_pmc.py DIR/f_results.db DIR/w_results.db
    python profiles/pmc_table.py cfg4 f64 DIR/f_results.db DIR/w_results.db      -> an entry of profiles/pmc_traffic.json"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import burgersutil  # noqa: E402


def workload(leg):
    if leg in ("headline", "cfg5") or leg.isdigit():
        n_f = {"headline": 10000, "cfg5": 1000000}.get(leg) or int(leg)
        np.random.seed(1234)
        r = burgersutil.prep_data(os.path.join(bench.PKG, "1d-burgers", "data", "burgers_shock.mat"), 100, n_f, noise=0.0)
        wl = bench.burgers_workload((r[9], r[7], r[8], r[11], r[10]))
        wl["w0"] = bench.canonical_weights()
        return wl
    return {"cfg3": bench.identification_workload, "cfg4": bench.schrodinger_workload}[leg]()


if __name__ == "__main__":
    leg = sys.argv[1] if len(sys.argv) > 1 else "headline"
    wl = workload(leg)
    for dtype in (sys.argv[2:] or ["f32", "f64"]):
        eng = bench.make_engine(dtype, 0, wl, 1, 0)
        eng.set_weights(wl["w0"])
        for _ in range(20):
            eng.loss_grad(want_grad=False)
        print(leg, dtype, "path", eng.kernel_path(), "points", wl["points"], "done", flush=True)
        eng.close()
