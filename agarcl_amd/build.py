"""Build libagarcl_hip.so (the product's only compute path) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "agar_engine.hip")
import glob
# every file under csrc/ is a dependency (one translation unit that #includes the .inl / .h files)
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(HERE, "..", "include", h) for h in ("agarcl_batch.h", "agarcl_vec.h")]
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


# (pellet slots per lane, all buckets visible, kind): one translation unit each; kind 0 = k_step / k_reset / k_respawn, 1 = k_quiet / k_fused
PARTS = [(ns, av, kind) for ns in (16, 32, 8, 4) for av in (1, 0) for kind in (0, 1)]
# the step units only: without machine-level loop-invariant code motion.  What it hoists in front of k_step's loops (lane masks, table
# constants, per-lane offsets: ~45 registers' worth) does not fit the 128-register budget and is spilled right there -- 11.5 KB of scratch
# written per arena-step.  Measured on MI355X (product flags / this, us per step): mode 6 at 4096 arenas 420.9 / 424.4, mid-game 78.1 / 76.6,
# C1 208.6 / 207.3; k_step's WRITE_SIZE per 4096-arena launch 57.1 -> 28.8 MB, scratch 460 -> 276 bytes per lane.  (The front kernels keep
# it: C2 at 4096 arenas 8.94 / 9.05.)
STEP_FLAGS = ["-mllvm", "-disable-machine-licm"]


def _compile(args):
    subprocess.check_call(args)
    return args[-1]


def build(force=False, verbose=False, extra=(), out=OUT, single=False, jobs=None):
    """Seventeen translation units compiled in parallel -- the main unit (-DAG_SPLIT_BUILD: host code, C ABI, observation kernels) and two
    part units per (NS, AV) pair (-DAG_PART_NS / -DAG_PART_AV / -DAG_PART_KIND: the step or the front kernels of that pair, agar_engine.hip
    "split build") -- linked into one shared library.  single=True compiles the same source as one unit (3-4 minutes instead of < 1).
    extra/out: diagnostic variants only (e.g. -DAGAR_PROFILE -> build_variants/lib_PROF.so, used by scripts/; build_variants/ is
    git-ignored and travels to the GPU box only while it exists -- delete it when the measurements are done)."""
    if not force and not needs_build() and out == OUT:
        return OUT
    base = [hipcc()] + FLAGS + list(extra)
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if single or os.environ.get("AGARCL_BUILD_SINGLE") == "1":   # (one unit: STEP_FLAGS would then reach the front kernels too, so they are left out)
        subprocess.check_call(base + ["-o", out, SRC])
        return out
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    comp = [f for f in base if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="agarcl_build_") as tmp:
        units = [comp + ["-DAG_SPLIT_BUILD", "-c", SRC, "-o", os.path.join(tmp, "main.o")]]
        for ns, av, kind in PARTS:
            units.append(comp + (STEP_FLAGS if kind == 0 else []) + ["-DAG_PART_NS=%d" % ns, "-DAG_PART_AV=%d" % av, "-DAG_PART_KIND=%d" % kind, "-c", SRC,
                                 "-o", os.path.join(tmp, "part_%d_%d_%d.o" % (ns, av, kind))])
        with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(_compile, units))
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def resource_table(path):
    """hipcc -Rpass-analysis=kernel-resource-usage over the SAME seventeen units with the SAME flags as build() (device code only), condensed
    into one line per kernel: profiles/rNN_kernel_resource_usage.txt"""
    import re
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    base = [hipcc()] + [f for f in FLAGS if f != "-shared"] + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", SRC]
    with tempfile.TemporaryDirectory(prefix="agarcl_res_") as tmp:
        units = [base + ["-DAG_SPLIT_BUILD", "-o", os.path.join(tmp, "main.o")]]
        for ns, av, kind in PARTS:
            units.append(base + (STEP_FLAGS if kind == 0 else []) + ["-DAG_PART_NS=%d" % ns, "-DAG_PART_AV=%d" % av, "-DAG_PART_KIND=%d" % kind,
                                                                     "-o", os.path.join(tmp, "p%d_%d_%d.o" % (ns, av, kind))])
        with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
            texts = list(pool.map(lambda u: subprocess.run(u, capture_output=True, text=True, check=True).stderr, units))
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    out = ["# hipcc -Rpass-analysis=kernel-resource-usage of agarcl_amd/csrc/agar_engine.hip, unit by unit with the flags of agarcl_amd/build.py "
           "(step units: %s), source sha %s" % (" ".join(STEP_FLAGS), h.hexdigest()[:16]),
           "# kernel | VGPRs | AGPRs | SGPRs | scratch bytes/lane | VGPR spills | SGPR spills | occupancy waves/SIMD | LDS bytes/block"]
    rows = {}
    for t in texts:
        for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
            g = lambda k: re.search(k + r": (\d+)", b).group(1)
            rows[b.split()[0]] = " | ".join([b.split()[0], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"),
                                              g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")])
    out += [rows[k] for k in sorted(rows)]
    open(path, "w").write("\n".join(out) + "\n")
    return len(rows)


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
    elif "--variant" in sys.argv:   # diagnostic builds:  build.py --variant NAME -DFOO [-DBAR ...]  ->  build_variants/lib_NAME.so
        os.makedirs(os.path.join(HERE, "..", "build_variants"), exist_ok=True)
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build(True, "-v" in sys.argv, [a for a in sys.argv if a.startswith("-D")], os.path.join(HERE, "..", "build_variants", "lib_%s.so" % name)))
    elif "--resources" in sys.argv:
        print(resource_table(sys.argv[sys.argv.index("--resources") + 1]), "kernels")
    elif "--pybind" in sys.argv:
        print(build_pybind(True))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, single="--single" in sys.argv))
