"""ctypes binding of libpinn_hip.so (C ABI: include/pinn_hip.h) -- the only way the Python
host reaches the GPU.  There is deliberately no CPU fallback: if the HIP library cannot be
loaded or no device is present, construction fails with an explicit error.

    from pinn_native import Engine, build
    eng = Engine(layers, lb, ub, pde="burgers", dtype="f32")
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
_REPO = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get("PINN_HIP_LIB") or os.path.join(_HERE, "libpinn_hip.so")
SOURCES = ["engine.hip", "fused20d_unit.hip", "fused20d_api.h", "fused20m_unit.hip", "fused20m_api.h", "kernels_generic.h", "kernels_fused20.h", "kernels_fused20m.h", "kernels_fused20d.h", "kernels_wide.h", "kernels_predict20.h",
           "kernels_disc.h", "kernels_sampling.h", "kernels_tile16.h", "kernels_tile16f.h", "kernels_xgmi.h", "kernels_optim.h", "wave.h"]
HEADER = os.path.join(_REPO, "include", "pinn_hip.h")

PDE_KINDS = {"burgers": 0, "burgers_ide": 1, "schrodinger": 2, "burgers_disc": 3, "burgers_disc_ide": 4}
DTYPES = {"f32": 0, "f64": 1, "float32": 0, "float64": 1}

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)


class PinnNativeError(RuntimeError):
    pass


COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC",
                "-mllvm", "-amdgpu-kernarg-preload-count=16"]   # leading argument dwords arrive in SGPRs (no s_load round trip)
UNITS = [("engine.hip", []), ("fused20d_unit.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]), ("fused20m_unit.hip", [])]


def _build_tag():
    """what a library was built from, besides file times: the flags (an ablation / stamps build shares the sources)"""
    return " ".join(COMMON_FLAGS) + " | " + " ; ".join("%s %s" % (u, " ".join(f)) for u, f in UNITS)


def _source_digest():
    """sha256 over the kernel sources and the C header, in SOURCES order: what a library was built from"""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(_CSRC, s) for s in SOURCES] + [HEADER]:
        if os.path.exists(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def library_digest(lib=None):
    """sources-sha256 recorded beside the library when it was built (what the code that RUNS was compiled from); None when
    the flags file is missing"""
    try:
        for line in open((lib or LIB_PATH) + ".flags"):
            if line.startswith("sources-sha256 "):
                return line.split()[1]
    except OSError:
        pass
    return None


def _stale(lib=None):
    """A library is current when the flags file beside it names these flags AND these source contents (a digest, not
    file times: a snapshot copied to another machine keeps no usable time order).  Without a flags file: file times."""
    lib = lib or LIB_PATH
    if not os.path.exists(lib):
        return True
    tag = lib + ".flags"
    if os.path.exists(tag):
        lines = open(tag).read().split("\n")
        digest = [l.split(" ", 1)[1] for l in lines if l.startswith("sources-sha256 ")]
        if lines[0] != _build_tag():
            return True
        if digest:
            return digest[0] != _source_digest()
    built = os.path.getmtime(lib)
    srcs = [os.path.join(_CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.exists(s) and os.path.getmtime(s) > built for s in srcs)


def build(force=False, verbose=False, stamps=False):
    """Compile the HIP engine for gfx950 into pinn_native/libpinn_hip.so (in-tree, so the
    shared object travels with the repo snapshot).  hipcc cross-compiles without a GPU.
    stamps=True builds the profiling variant libpinn_hip_stamps.so (-DPINN_STAMPS) instead.
    Safe against concurrent callers (several ranks importing at once): a file lock serialises them, objects go to
    a private temporary directory, the library is moved into place atomically."""
    import fcntl
    import tempfile
    out = LIB_PATH.replace(".so", "_stamps.so") if stamps else LIB_PATH
    if not force and not stamps and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise PinnNativeError("hipcc not found; cannot build libpinn_hip.so")
    try:
        lock = open(os.path.join(_HERE, ".build.lock"), "w")
    except OSError as e:                          # a read-only install: nothing can be built in place
        raise PinnNativeError("cannot build in %s (%s); build the library where the tree is writable" % (_HERE, e))
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stamps and not _stale():        # another process built it while we waited
            return LIB_PATH
        # three translation units (compiled concurrently), one shared object: k_fused20d wants its matrix results in
        # VGPRs (csrc/fused20d_api.h), every other kernel keeps hipcc's default allocation; fused20m_unit.hip holds
        # the float32 register-stash kernel at the depths other than 8
        common = [hipcc] + COMMON_FLAGS + (["-DPINN_STAMPS"] if stamps else [])
        digest = _source_digest()
        with tempfile.TemporaryDirectory(prefix="pinn_build_", dir=_HERE) as tmp:
            objs, procs = [], []
            for src, extra in UNITS:
                obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
                cmd = common + extra + ["-c", os.path.join(_CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
                objs.append(obj)
            failures = []
            for cmd, pr in procs:                              # wait for ALL compiles before reporting
                log = pr.communicate()[0]
                if pr.returncode != 0:
                    failures.append("hipcc failed (%s):\n%s" % (" ".join(cmd), log))
            if failures:
                raise PinnNativeError("\n".join(failures))
            tmp_so = os.path.join(tmp, "lib.so")
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp_so, "-lrccl"]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise PinnNativeError("hipcc link failed:\n" + res.stdout + res.stderr)
            os.replace(tmp_so, out)
            for obj in objs:                                   # kept for the ablation builds of profiles/ (relinked there)
                os.replace(obj, os.path.join(_HERE, os.path.basename(obj).replace(".o", "_stamps.o" if stamps else ".o")))
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        with open(out + ".flags", "w") as fh:                  # flags + the compiler the kernels were validated with
            fh.write(_build_tag() + ("\n-DPINN_STAMPS" if stamps else "") + "\nsources-sha256 " + digest + "\n" +
                     "\n".join(ver[:3]) + "\n")
    return out


_LIB = None

_SIGNATURES = {
    "pinn_last_error": (ctypes.c_char_p, []),
    "pinn_abi_version": (ctypes.c_int, []),
    "pinn_device_count": (ctypes.c_int, [_c_int_p]),
    "pinn_runtime_versions": (ctypes.c_int, [_c_int_p, _c_int_p, _c_int_p]),
    "pinn_device_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, _c_int_p,
                                        ctypes.POINTER(ctypes.c_int64)]),
    "pinn_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _c_int_p, ctypes.c_int,
                                   _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int]),
    "pinn_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pinn_num_params": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "pinn_set_collocation": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64,
                                            ctypes.c_int64]),
    "pinn_set_data": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.c_int64,
                                     ctypes.c_int64]),
    "pinn_set_boundary": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p,
                                         ctypes.c_int64, ctypes.c_int64]),
    "pinn_set_pde_params": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int]),
    "pinn_set_weights": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    "pinn_get_weights": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    "pinn_loss_grad": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    "pinn_adam_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double]),
    "pinn_adam_run_terms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_double_p]),
    "pinn_adam_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_double_p]),
    "pinn_lbfgs_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double]),
    "pinn_lbfgs_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int_p, _c_double_p,
                                      _c_int_p, _c_int_p]),
    "pinn_adam_enqueue": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int_p]),
    "pinn_adam_enqueue_terms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int_p]),
    "pinn_adam_collect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_double_p]),
    "pinn_lbfgs_enqueue": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_int_p]),
    "pinn_lbfgs_collect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _c_int_p, _c_double_p,
                                          _c_int_p, _c_int_p]),
    "pinn_weights_snapshot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pinn_weights_restore": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pinn_lbfgs_get_x": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    "pinn_lbfgs_set_mode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pinn_predict": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, _c_double_p]),
    "pinn_error_l2": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int,
                                     _c_double_p]),
    "pinn_get_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64)]),
    "pinn_residual": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    "pinn_residual_at": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, _c_double_p]),
    "pinn_comm_xgmi_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    "pinn_comm_xgmi_attach": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, _c_int_p]),
    "pinn_comm_xgmi_selftest": (ctypes.c_int, [ctypes.c_void_p, _c_int_p]),
    "pinn_comm_benchmark": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "pinn_comm_set_mode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pinn_comm_get_mode": (ctypes.c_int, [ctypes.c_void_p, _c_int_p]),
    "pinn_lhs_collocation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_uint64]),
    "pinn_get_collocation": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64]),
    "pinn_disc_set_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_double_p, _c_double_p,
                                           ctypes.c_int64, _c_double_p, ctypes.c_int]),
    "pinn_disc_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_double_p, ctypes.c_int64,
                                         _c_double_p]),
    "pinn_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "pinn_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.c_int]),
    "pinn_timing_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "pinn_timing_read": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_int_p]),
    "pinn_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "pinn_set_kernel_path": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pinn_get_kernel_path": (ctypes.c_int, [ctypes.c_void_p, _c_int_p]),
    "pinn_debug_coef_stamps": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong)]),
    "pinn_debug_t16f_stamps": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong)]),
    "pinn_debug_t16_deal": (ctypes.c_int, [ctypes.c_int, _c_int_p]),
    "pinn_debug_stamps": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong),
                                         ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def _torch_lib_dir():
    """torch/lib of the installed PyTorch-ROCm wheel, found WITHOUT importing torch; None when torch is absent"""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def _bind_runtime():
    """Decide which HIP runtime + RCCL this process runs on BEFORE the engine is mapped.

    The image holds two sets with the same sonames (libamdhip64.so.7, librccl.so.1, libhsa-runtime64.so.1): /opt/rocm
    (ROCm 7.2, the toolchain the engine is compiled with; the engine's RUNPATH) and the set bundled in torch/lib
    (ROCm 7.0).  The dynamic linker gives the engine whichever is mapped first, and torch maps its own copies by path
    regardless -- an engine that bound /opt/rocm in a process that later imports torch leaves TWO HIP runtimes in one
    process (measured here: heap corruption at exit).  So a process that needs torch.distributed (every rank of a
    data-parallel launch) must bind torch's set, and it must do so before the engine is mapped.
    PINN_HIP_RUNTIME = auto (default): torch's set when this is one rank of WORLD_SIZE > 1 or torch is already
                                       imported, /opt/rocm otherwise (a plain single-process run never needs torch);
                       torch: always torch's set (torch is imported first) -- the same runtime at N = 1 as at N = 8;
                       rocm:  /opt/rocm; refuses to proceed when torch is already in the process.
    runtime_info() reports what was bound; bench.py prints it in every line."""
    import sys
    policy = os.environ.get("PINN_HIP_RUNTIME", "auto").lower()
    if policy not in ("auto", "torch", "rocm"):
        raise PinnNativeError("PINN_HIP_RUNTIME must be auto, torch or rocm (got %r)" % policy)
    have_torch = "torch" in sys.modules
    if policy == "rocm":
        if have_torch:
            raise PinnNativeError("PINN_HIP_RUNTIME=rocm, but torch (with its bundled HIP runtime) is already imported "
                                  "in this process: two HIP runtimes would be mapped")
        return "rocm"
    if policy == "auto" and not have_torch and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return "rocm"
    if have_torch:
        return "torch"
    if _torch_lib_dir() is None:
        if policy == "torch":
            raise PinnNativeError("PINN_HIP_RUNTIME=torch, but no PyTorch-ROCm wheel with a bundled runtime is installed")
        return "rocm"
    # import torch itself, not just its libraries: torch's librccl mapped by path BEFORE `import torch` aborts at exit
    # ("double free or corruption", measured in this image even with no engine in the process), and so does the
    # engine-then-torch order under either set.  torch first, engine second is the one order that is clean.
    import torch.distributed  # noqa: F401
    return "torch"


def runtime_info():
    """-> {"hip_runtime", "hip_driver", "rccl_version", "libamdhip64_path", "librccl_path", "libhsa_path", "bound"}: the
    versions the engine's own calls report (pinn_runtime_versions) and the files this process mapped for them"""
    lib = load()
    v = [ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)]
    if lib.pinn_runtime_versions(ctypes.byref(v[0]), ctypes.byref(v[1]), ctypes.byref(v[2])) != 0:
        raise PinnNativeError(lib.pinn_last_error().decode())
    paths = {"libamdhip64": [], "librccl": [], "libhsa-runtime64": []}
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                f = line.split()
                if len(f) >= 6 and "x" in f[1]:
                    for key in paths:
                        if os.path.basename(f[5]).startswith(key) and f[5] not in paths[key]:
                            paths[key].append(f[5])
    except OSError:
        pass
    one = lambda k: paths[k][0] if len(paths[k]) == 1 else (paths[k] or None)     # a list = more than one copy mapped (a fault)
    r = v[2].value
    return {"hip_runtime": v[0].value, "hip_driver": v[1].value,
            "rccl_version": "%d.%d.%d" % (r // 10000, (r // 100) % 100, r % 100), "rccl_version_code": r,
            "libamdhip64_path": one("libamdhip64"), "librccl_path": one("librccl"), "libhsa_path": one("libhsa-runtime64"),
            "bound": "torch" if (paths["libamdhip64"] and "/torch/lib/" in paths["libamdhip64"][0]) else "rocm",
            "single_runtime": all(len(paths[k]) <= 1 for k in paths)}


def load():
    """dlopen the engine (building it first if the sources are newer and hipcc exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.environ.get("PINN_HIP_LIB") and _stale():     # an explicitly named library (a variant build) is used as it is
        try:
            build()
        except PinnNativeError as e:
            if not os.path.exists(LIB_PATH):
                raise
            import warnings
            warnings.warn("libpinn_hip.so is older than its sources and could not be rebuilt (%s): using the stale "
                          "library" % str(e).splitlines()[0], RuntimeWarning)
    if not os.path.exists(LIB_PATH):
        raise PinnNativeError(
            "libpinn_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; "
            "g.build()'`.  There is no CPU fallback." % LIB_PATH)
    _bind_runtime()
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def device_count():
    n = ctypes.c_int(0)
    lib = load()
    if lib.pinn_device_count(ctypes.byref(n)) != 0:
        return 0
    return n.value


def device_info(device=0):
    lib = load()
    buf = ctypes.create_string_buffer(256)
    ncu = ctypes.c_int(0)
    mem = ctypes.c_int64(0)
    if lib.pinn_device_info(device, buf, 256, ctypes.byref(ncu), ctypes.byref(mem)) != 0:
        raise PinnNativeError(lib.pinn_last_error().decode())
    return {"name": buf.value.decode(), "compute_units": ncu.value, "hbm_bytes": mem.value}


class Engine(object):
    """One engine context = one GPU, one stream, one network + training set."""

    def __init__(self, layers, lb, ub, pde="burgers", dtype="f32", device=0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        if pde not in PDE_KINDS:
            raise ValueError("pde must be one of %s" % sorted(PDE_KINDS))
        if dtype not in DTYPES:
            raise ValueError("dtype must be one of %s" % sorted(DTYPES))
        self.layers = [int(v) for v in layers]
        self.pde, self.dtype = pde, ("f64" if DTYPES[dtype] else "f32")
        self.n_out = self.layers[-1]
        arr = (ctypes.c_int * len(self.layers))(*self.layers)
        self.n_in = self.layers[0]
        lb, ub = _f64(lb, (self.n_in,)), _f64(ub, (self.n_in,))
        self._check(self._lib.pinn_create(ctypes.byref(self._h), arr, len(self.layers), _dp(lb),
                                          _dp(ub), PDE_KINDS[pde], DTYPES[dtype], int(device)))
        n = ctypes.c_int64(0)
        self._check(self._lib.pinn_num_params(self._h, ctypes.byref(n)))
        self.n_params = n.value
        self.n_f = self.n_u = self.n_b = 0

    def _check(self, rc):
        if rc != 0:
            raise PinnNativeError("libpinn_hip: %s (code %d)" % (
                self._lib.pinn_last_error().decode(errors="replace"), rc))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pinn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- point sets ----------------------------------------------------------------------
    def set_collocation(self, X_f, n_total=None):
        X_f = _f64(X_f).reshape(-1, 2)
        self.n_f = X_f.shape[0]
        self._check(self._lib.pinn_set_collocation(self._h, _dp(X_f), self.n_f,
                                                   self.n_f if n_total is None else int(n_total)))

    def lhs_collocation(self, n_design, seed, first=0, count=None):
        """Draw collocation points [first, first+count) of an n_design-point Latin hypercube on the device."""
        count = int(n_design) - int(first) if count is None else int(count)
        self._check(self._lib.pinn_lhs_collocation(self._h, int(n_design), int(first), count, int(seed)))
        self.n_f = count

    def get_collocation(self):
        X = np.empty((self.n_f, 2), dtype=np.float64)
        self._check(self._lib.pinn_get_collocation(self._h, _dp(X), self.n_f))
        return X

    def set_data(self, X_u, u, n_total=None):
        X_u = _f64(X_u).reshape(-1, 2)
        u = _f64(u).reshape(X_u.shape[0], self.n_out)
        self.n_u = X_u.shape[0]
        self._check(self._lib.pinn_set_data(self._h, _dp(X_u), _dp(u), self.n_u,
                                            self.n_u if n_total is None else int(n_total)))

    def set_boundary(self, X_lb, X_ub, n_total=None):
        X_lb, X_ub = _f64(X_lb).reshape(-1, 2), _f64(X_ub).reshape(-1, 2)
        if X_lb.shape != X_ub.shape:
            raise ValueError("X_lb and X_ub must pair up")
        self.n_b = X_lb.shape[0]
        self._check(self._lib.pinn_set_boundary(self._h, _dp(X_lb), _dp(X_ub), self.n_b,
                                                self.n_b if n_total is None else int(n_total)))

    def set_pde_params(self, *p):
        p = _f64(p)
        self._check(self._lib.pinn_set_pde_params(self._h, _dp(p), p.size))

    # ---- weights ---------------------------------------------------------------------------
    def set_weights(self, w):
        w = _f64(w).ravel()
        self._check(self._lib.pinn_set_weights(self._h, _dp(w), w.size))

    def get_weights(self):
        w = np.empty(self.n_params, dtype=np.float64)
        self._check(self._lib.pinn_get_weights(self._h, _dp(w), w.size))
        return w

    # ---- evaluation ------------------------------------------------------------------------
    def loss_grad(self, want_grad=True):
        loss = ctypes.c_double(0.0)
        terms = np.zeros(3, dtype=np.float64)
        grad = np.empty(self.n_params, dtype=np.float64) if want_grad else None
        self._check(self._lib.pinn_loss_grad(self._h, ctypes.byref(loss),
                                             _dp(grad) if want_grad else None, _dp(terms)))
        return loss.value, grad, terms

    # ---- discrete-time models (pde "burgers_disc", "burgers_disc_ide") -------------------------
    def disc_set_stage(self, stage, x, target, M=None):
        """Stage set `stage` (0/1): points x [n], targets [n] (broadcast over the outputs), M [n_out, q] =
        step-scaled IRK table or None (no IRK term).  See include/pinn_hip.h."""
        x, target = _f64(x).ravel(), _f64(target).ravel()
        if x.size != target.size:
            raise ValueError("x and target must pair up")
        q = 0
        if M is not None:
            M = _f64(M)
            if M.ndim != 2 or M.shape[0] != self.n_out:
                raise ValueError("M must be [n_out, q]")
            q = M.shape[1]
        self._check(self._lib.pinn_disc_set_stage(self._h, int(stage), _dp(x), _dp(target), x.size,
                                                  _dp(M) if M is not None else None, q))

    def disc_predict(self, stage, x):
        x = _f64(x).ravel()
        out = np.empty((x.size, self.n_out), dtype=np.float64)
        self._check(self._lib.pinn_disc_predict(self._h, int(stage), _dp(x), x.size, _dp(out)))
        return out

    def predict(self, X):
        X = _f64(X).reshape(-1, self.n_in)
        out = np.empty((X.shape[0], self.n_out), dtype=np.float64)
        self._check(self._lib.pinn_predict(self._h, _dp(X), X.shape[0], _dp(out)))
        return out

    def error_l2(self, X, ref, modulus=False):
        """||ref - model(X)||_2 / ||ref||_2 reduced on the device (the scripts' error metric).  modulus=True compares
        |h| = sqrt(sum_o out_o^2) with ref [n] (Schrodinger)."""
        X = _f64(X).reshape(-1, self.n_in)
        ref = _f64(ref).reshape(-1) if modulus else _f64(ref).reshape(X.shape[0], self.n_out)
        if ref.shape[0] != X.shape[0]:
            raise ValueError("ref must hold one row per point")
        err = ctypes.c_double(0.0)
        self._check(self._lib.pinn_error_l2(self._h, _dp(X), _dp(ref), X.shape[0], 1 if modulus else 0,
                                            ctypes.byref(err)))
        return err.value

    def status(self):
        """(loss+grad evaluations so far, 1-based number of the first one with a non-finite loss or 0)"""
        n, bad = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.pinn_get_status(self._h, ctypes.byref(n), ctypes.byref(bad)))
        return n.value, bad.value

    def residual(self):
        n = self.n_u if self.pde == "burgers_ide" else self.n_f
        f = np.empty((n, self.n_out), dtype=np.float64)
        self._check(self._lib.pinn_residual(self._h, _dp(f), n))
        return f

    def residual_at(self, X):
        """f_model at arbitrary points X [n, 2] -> [n, n_out]"""
        X = _f64(X).reshape(-1, 2)
        f = np.empty((X.shape[0], self.n_out), dtype=np.float64)
        self._check(self._lib.pinn_residual_at(self._h, _dp(X), X.shape[0], _dp(f)))
        return f

    # ---- optimisers ------------------------------------------------------------------------
    def adam_init(self, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self._check(self._lib.pinn_adam_init(self._h, lr, beta1, beta2, eps))

    def adam_run_terms(self, n_steps):
        """[n_steps, 3] = (residual, data, boundary) loss parts before each update."""
        terms = np.empty((max(int(n_steps), 1), 3), dtype=np.float64)
        self._check(self._lib.pinn_adam_run_terms(self._h, int(n_steps), _dp(terms)))
        return terms[:n_steps]

    def adam_run(self, n_steps, want_losses=True):
        if not want_losses:
            self._check(self._lib.pinn_adam_run(self._h, int(n_steps), None))
            return None
        losses = np.empty(max(int(n_steps), 1), dtype=np.float64)
        self._check(self._lib.pinn_adam_run(self._h, int(n_steps), _dp(losses)))
        return losses[:n_steps]

    # -- the same loops with the host one chunk behind the GPU (include/pinn_hip.h: pinn_*_enqueue / _collect) --------
    MAX_IN_FLIGHT = 4

    def adam_enqueue(self, n_steps, terms=False):
        """n_steps Adam iterations into the stream -> ticket (returns at once); adam_collect(ticket) -> their losses
        ([n] sums, or with terms=True the [n, 3] parts (residual, data, boundary) of adam_run_terms)"""
        t = ctypes.c_int(0)
        fn = self._lib.pinn_adam_enqueue_terms if terms else self._lib.pinn_adam_enqueue
        self._check(fn(self._h, int(n_steps), ctypes.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        self._tickets[t.value] = -int(n_steps) if terms else int(n_steps)      # (negative: a terms chunk)
        return t.value

    def adam_collect(self, ticket):
        n = self._tickets[ticket]
        terms, n = n < 0, abs(n)
        losses = np.empty((max(n, 1), 3) if terms else max(n, 1), dtype=np.float64)
        self._check(self._lib.pinn_adam_collect(self._h, int(ticket), _dp(losses)))
        del self._tickets[ticket]
        return losses[:n]

    def lbfgs_enqueue(self, n_iters):
        t = ctypes.c_int(0)
        self._check(self._lib.pinn_lbfgs_enqueue(self._h, int(n_iters), ctypes.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        self._lb_uncollected = getattr(self, "_lb_uncollected", 0) + int(n_iters)
        self._tickets[t.value] = int(n_iters)
        return t.value

    def lbfgs_collect(self, ticket):
        """-> (iters, losses, done) of the log entries that became due with this chunk"""
        cap = self._lb_uncollected + 1            # every iteration enqueued since the last collect may have logged one entry
        iters = np.zeros(cap, dtype=np.int32)
        losses = np.zeros(cap, dtype=np.float64)
        n_logged, done = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.pinn_lbfgs_collect(self._h, int(ticket), cap, iters.ctypes.data_as(_c_int_p), _dp(losses),
                                                 ctypes.byref(n_logged), ctypes.byref(done)))
        del self._tickets[ticket]
        self._lb_uncollected = sum(abs(v) for v in self._tickets.values())
        k = n_logged.value
        return iters[:k].copy(), losses[:k].copy(), done.value

    N_SNAPSHOTS = 4

    def weights_snapshot(self, slot):
        """device-side copy of the weights into slot 0..3, in stream order (no synchronisation)"""
        self._check(self._lib.pinn_weights_snapshot(self._h, int(slot)))

    def weights_restore(self, slot):
        self._check(self._lib.pinn_weights_restore(self._h, int(slot)))

    def lbfgs_begin(self, max_iter, lr, n_corr, tol_fun, tol_x=1e-19, max_eval=0.0):
        self._tickets, self._lb_uncollected = {}, 0          # chunks still in flight are dropped by the engine (a restart)
        self._lb_max_iter = int(max_iter)
        self._check(self._lib.pinn_lbfgs_begin(self._h, int(max_iter), lr, int(n_corr), tol_fun,
                                               tol_x, max_eval))

    def lbfgs_set_mode(self, mode):
        self._check(self._lib.pinn_lbfgs_set_mode(self._h, int(mode)))

    def lbfgs_run(self, n_iters):
        cap = max(int(n_iters), 1)
        iters = np.zeros(cap, dtype=np.int32)
        losses = np.zeros(cap, dtype=np.float64)
        n_logged, done = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.pinn_lbfgs_run(self._h, int(n_iters),
                                             iters.ctypes.data_as(_c_int_p), _dp(losses),
                                             ctypes.byref(n_logged), ctypes.byref(done)))
        k = n_logged.value
        return iters[:k].copy(), losses[:k].copy(), done.value

    def lbfgs_x(self):
        x = np.empty(self.n_params, dtype=np.float64)
        self._check(self._lib.pinn_lbfgs_get_x(self._h, _dp(x), x.size))
        return x

    # ---- multi-GPU -------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = load()
        buf = ctypes.create_string_buffer(128)
        if lib.pinn_comm_unique_id(buf) != 0:
            raise PinnNativeError(lib.pinn_last_error().decode())
        return buf.raw

    def comm_init(self, unique_id, n_ranks, rank):
        self._check(self._lib.pinn_comm_init(self._h, bytes(unique_id), int(n_ranks), int(rank)))

    # ---- measurement -----------------------------------------------------------------------
    def comm_xgmi_export(self, n_ranks, rank):
        buf = ctypes.create_string_buffer(64)
        self._check(self._lib.pinn_comm_xgmi_export(self._h, int(n_ranks), int(rank), buf))
        return buf.raw

    def comm_xgmi_attach(self, handles):
        """handles: list of the 64-byte handles of all ranks, rank order.  True if every peer mailbox got mapped."""
        blob = b"".join(handles)
        ok = ctypes.c_int(0)
        self._check(self._lib.pinn_comm_xgmi_attach(self._h, blob, len(handles), ctypes.byref(ok)))
        return bool(ok.value)

    def comm_xgmi_selftest(self):
        ok = ctypes.c_int(0)
        self._check(self._lib.pinn_comm_xgmi_selftest(self._h, ctypes.byref(ok)))
        return bool(ok.value)

    def comm_benchmark(self, mode, iters=200):
        us = ctypes.c_double(0.0)
        self._check(self._lib.pinn_comm_benchmark(self._h, {"rccl": 1, "mailbox": 2}.get(mode, mode), int(iters),
                                                  ctypes.byref(us)))
        return us.value

    def comm_set_mode(self, mode):
        self._check(self._lib.pinn_comm_set_mode(self._h, {"rccl": 1, "mailbox": 2}.get(mode, mode)))

    def comm_mode(self):
        m = ctypes.c_int(0)
        self._check(self._lib.pinn_comm_get_mode(self._h, ctypes.byref(m)))
        return {0: "none", 1: "rccl", 2: "mailbox"}[m.value]

    def timing_enable(self, max_evals, every=1):
        self._check(self._lib.pinn_timing_enable(self._h, int(max_evals), int(every)))

    def timing_read(self):
        ms = np.zeros(5, dtype=np.float64)
        n = ctypes.c_int(0)
        self._check(self._lib.pinn_timing_read(self._h, _dp(ms), ctypes.byref(n)))
        return {"fwd_ms": ms[0], "sweeps_ms": ms[1], "eval_ms": ms[2], "empty_bracket_ms": ms[3],
                "kernel_exact": bool(ms[4]), "n": n.value}

    def sync(self):
        self._check(self._lib.pinn_sync(self._h))

    def set_kernel_path(self, path):
        self._check(self._lib.pinn_set_kernel_path(self._h, int(path)))

    def debug_stamps(self):
        """(profiling build) -> int64 array [n_waves, 32] of s_memtime ticks for one evaluation"""
        n_wg = (2 * self.n_b + self.n_u + self.n_f + 63) // 64
        buf = np.zeros((n_wg * 4, 32), dtype=np.int64)
        n = ctypes.c_int64(0)
        self._check(self._lib.pinn_debug_stamps(
            self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), buf.size, ctypes.byref(n)))
        return buf[:n.value]

    def kernel_path(self):
        p = ctypes.c_int(0)
        self._check(self._lib.pinn_get_kernel_path(self._h, ctypes.byref(p)))
        return p.value
