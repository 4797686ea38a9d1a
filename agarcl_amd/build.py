"""Build libagarcl_hip.so (the product's only compute path) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "agar_engine.hip")
import glob
# every file under csrc/ is a dependency (one translation unit that #includes the .inl / .h files)
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(HERE, "..", "include", "agarcl_batch.h")]
OUT = os.path.join(HERE, "libagarcl_hip.so")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # bit-exact fp32 vs the reference's x86-64 build: no FMA contraction, IEEE divide / sqrt
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=(), out=OUT):
    """extra/out: diagnostic variants only (e.g. -DAGAR_PROFILE -> libagarcl_hip_prof.so, used by scripts/)."""
    if not force and not needs_build() and out == OUT:
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + ["-o", out, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--profile" in sys.argv:
        print(build(True, "-v" in sys.argv, ["-DAGAR_PROFILE"], os.path.join(HERE, "libagarcl_hip_prof.so")))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
