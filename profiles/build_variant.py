"""Build a variant of libpinn_hip.so for a same-box A/B: the three translation units compiled with extra flags, linked
into pinns-tf2.0_amd/pinn_native/abl/libpinn_hip_<name>.so (git-ignored; shipped to the GPU box; selected with
PINN_HIP_LIB=<path>).
    python profiles/build_variant.py <name> [extra hipcc flags ...]      e.g.  preload -mllvm -amdgpu-kernarg-preload-count=16
PINN_VARIANT_CSRC=<dir>: compile that copy of csrc/ instead (e.g. `git archive <rev> pinns-tf2.0_amd/csrc include | tar -x -C /tmp/old`
for an A/B against an earlier revision of a kernel)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pinns-tf2.0_amd"))
import pinn_native as pn

name, extra = sys.argv[1], sys.argv[2:]
csrc = os.environ.get("PINN_VARIANT_CSRC") or os.path.join(ROOT, "pinns-tf2.0_amd", "csrc")
out_dir = os.path.join(ROOT, "pinns-tf2.0_amd", "pinn_native", "abl")
os.makedirs(out_dir, exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    procs, objs = [], []
    for src, unit_flags in pn.UNITS:
        obj = os.path.join(tmp, src.replace(".hip", ".o"))
        cmd = ["hipcc"] + pn.COMMON_FLAGS + unit_flags + extra + ["-c", os.path.join(csrc, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, p in procs:
        log = p.communicate()[0]
        if p.returncode:
            sys.exit("failed: %s\n%s" % (" ".join(cmd), log))
    out = os.path.join(out_dir, "libpinn_hip_%s.so" % name)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-lrccl"])
print(out)
