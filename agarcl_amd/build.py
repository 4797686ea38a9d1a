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


PARTS = [(ns, av) for ns in (4, 8, 16, 32) for av in (1, 0)]   # (pellet slots per lane, all buckets visible): one translation unit each


def _compile(args):
    subprocess.check_call(args)
    return args[-1]


def build(force=False, verbose=False, extra=(), out=OUT, single=False, jobs=None):
    """Nine translation units compiled in parallel -- the main unit (-DAG_SPLIT_BUILD: host code, C ABI, observation kernels) and one
    part unit per (NS, AV) pair (-DAG_PART_NS / -DAG_PART_AV: the step / reset kernels of that pair, agar_engine.hip "split build") --
    linked into one shared library.  single=True compiles the same source as one unit (3-4 minutes instead of < 1).
    extra/out: diagnostic variants only (e.g. -DAGAR_PROFILE -> build_variants/lib_PROF.so, used by scripts/; build_variants/ is
    git-ignored and travels to the GPU box only while it exists -- delete it when the measurements are done)."""
    if not force and not needs_build() and out == OUT:
        return OUT
    base = [hipcc()] + FLAGS + list(extra)
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if single or os.environ.get("AGARCL_BUILD_SINGLE") == "1":
        subprocess.check_call(base + ["-o", out, SRC])
        return out
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    comp = [f for f in base if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="agarcl_build_") as tmp:
        units = [comp + ["-DAG_SPLIT_BUILD", "-c", SRC, "-o", os.path.join(tmp, "main.o")]]
        for ns, av in PARTS:
            units.append(comp + ["-DAG_PART_NS=%d" % ns, "-DAG_PART_AV=%d" % av, "-c", SRC, "-o", os.path.join(tmp, "part_%d_%d.o" % (ns, av))])
        with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(_compile, units))
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def pybind_out():
    import sysconfig
    return os.path.join(HERE, "..", "agarcl" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False):
    """g++ -> <repo>/agarcl.<abi>.so: the reference's pybind11 module name (`import agarcl`) over the C ABI of libagarcl_hip.so
    (csrc/agarcl_pybind.cpp: host-only, links the HIP library, rpath $ORIGIN/agarcl_amd)."""
    import sysconfig
    import pybind11
    src, out = os.path.join(HERE, "csrc", "agarcl_pybind.cpp"), os.path.abspath(pybind_out())
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "agarcl_batch.h"))):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           src, "-o", out, "-L" + HERE, "-l:libagarcl_hip.so", "-Wl,-rpath,$ORIGIN/agarcl_amd"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--profile" in sys.argv:
        os.makedirs(os.path.join(HERE, "..", "build_variants"), exist_ok=True)
        print(build(True, "-v" in sys.argv, ["-DAGAR_PROFILE"], os.path.join(HERE, "..", "build_variants", "lib_PROF.so")))
    elif "--pybind" in sys.argv:
        print(build_pybind(True))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, single="--single" in sys.argv))
