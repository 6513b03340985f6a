"""CPU oracle: numpy float64 restatement of the reference's hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is the *checker* for the HIP engine,
never the thing shipped or measured: only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  The product path
(`pinns-tf2.0_amd/`) never imports `oracle` and fails loudly if the HIP library
is missing.

What it restates (reference = pierremtb/PINNs-TF2.0, paths relative to its root):
  mlp.py      utils/neuralnetwork.py:24-47 (model), :68-89 (flat layout), Taylor-mode
              forward + hand reverse sweep equivalent to the nested GradientTapes of
              1d-burgers/inf_cont_burgers.py:65-90
  pde.py      1d-burgers/inf_cont_burgers.py:59-98, 1d-burgers/ide_cont_burgers.py:52-118,
              1dcomplex-schrodinger/inf_cont_schrodinger.py:47-135
  optim.py    utils/neuralnetwork.py:19-22,105-116 (Adam loop; TF-2.0 ResourceApplyAdam
              formula), utils/custom_lbfgs.py:39-236 (L-BFGS, incl. the last-iteration quirk)
  init.py     Keras glorot_normal as restated by tests/ref_shims/tensorflow.py

Parity pinning: TensorFlow itself (requirements.txt:5, un-vendored, not installable
here) cannot be run, and the reference holds no golden vectors of its own.  The oracle
is therefore pinned against the reference's *own Python sources* executed unmodified
over tests/ref_shims (torch-f64 stand-in for the tensorflow module): see
tests/golden/make_golden.py and tests/test_oracle_vs_golden.py.  At the TensorFlow
boundary itself (matmul/tanh/autodiff/Adam kernels) parity is unpinned upstream.
"""
