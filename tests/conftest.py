import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pinns-tf2.0_amd")
for p in (ROOT, PKG, os.path.join(PKG, "utils"), os.path.join(PKG, "1d-burgers"),
          os.path.join(PKG, "1dcomplex-schrodinger")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BURGERS_MAT = os.path.join(PKG, "1d-burgers", "data", "burgers_shock.mat")
NLS_MAT = os.path.join(PKG, "1dcomplex-schrodinger", "data", "NLS.mat")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def burgers_sets():
    """{(N_u, N_f): prep_data tuple} for the seeds the goldens were generated with."""
    import burgersutil
    cache = {}

    def get(N_u, N_f):
        key = (N_u, N_f)
        if key not in cache:
            np.random.seed(1234)
            cache[key] = burgersutil.prep_data(BURGERS_MAT, N_u, N_f, noise=0.0)
        return cache[key]
    return get


@pytest.fixture(scope="session")
def schrodinger_sets():
    import schrodingerutil
    cache = {}

    def get(N_0, N_b, N_f):
        key = (N_0, N_b, N_f)
        if key not in cache:
            np.random.seed(1234)
            cache[key] = schrodingerutil.prep_data(NLS_MAT, N_0, N_b, N_f, noise=0.0)
        return cache[key]
    return get


_MEASURED = os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl")


@pytest.fixture
def record(request):
    """record(name=value, ...): appends the deviations a GPU parity test measured to gpurun_out/parity_measured.jsonl
    (scratch; a copy of the last full run is committed under profiles/), next to asserting them."""
    import json

    def rec(**kw):
        try:
            os.makedirs(os.path.dirname(_MEASURED), exist_ok=True)
            with open(_MEASURED, "a") as fh:
                fh.write(json.dumps({"test": request.node.name, **{k: (float(v) if hasattr(v, "__float__") else v)
                                                                    for k, v in kw.items()}}) + "\n")
        except OSError:
            pass
    return rec


def ensemble_accepts(errors, value):
    """Acceptance rule for a final error of a roundoff-chaotic schedule, given the reference's own ensemble `errors`
    (final errors of >= 25 reference runs whose initial weights differ by a few ulp): the value must lie INSIDE the
    range the reference itself produced, [min, max] -- no margin.  -> (ok, min, max)
    (Rounds 1-2 accepted median +- the most distant member, i.e. values outside the observed range.)"""
    e = np.sort(np.asarray(errors, dtype=np.float64))
    assert e.size >= 25, "the acceptance range is only meaningful over a sizeable ensemble (make_band.py)"
    return bool(e[0] <= value <= e[-1]), float(e[0]), float(e[-1])


def same_distribution_p(a, b):
    """two-sided Mann-Whitney U (rank) test that two samples of final errors come from one distribution -> p-value.
    What is asserted is p >= 1e-3: a correct implementation fails one run in a thousand, a biased one (all its errors
    below / above the reference's) gives p ~ 1e-9 at 25 + 25 members."""
    from scipy.stats import mannwhitneyu
    return float(mannwhitneyu(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), alternative="two-sided").pvalue)
