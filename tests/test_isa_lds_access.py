"""The step kernels address LDS with ds_* instructions only.  A flat_* access picks its aperture from the address register before the
instruction's immediate offset is added, and hipcc folds constant index parts into that immediate: an LDS access through a generic
pointer whose register part drops below the wave's LDS block faulted the queue in round 2 (the fix hands LDS over
as an offset into ag_lds).  This test disassembles the built library and keeps flat_* memory instructions out of the step path."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "agarcl_amd", "libagarcl_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def flat_counts(tmp_path_factory):
    for tool in ("clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip("ROCm LLVM tools not available")
    if not os.path.exists(SO) or not shutil.which("objcopy"):
        pytest.skip("library not built")
    d = str(tmp_path_factory.mktemp("isa"))
    fat = os.path.join(d, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", SO, fat])
    # the library is linked from several translation units (agarcl_amd/build.py): the section holds one offload bundle per unit
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in the library"
    counts = {}   # function -> [flat loads, flat stores + atomics]
    head = re.compile(r"^[0-9a-f]+ <(.*)>:$")
    for k, (b0, b1) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        one, co = os.path.join(d, "bundle%d" % k), os.path.join(d, "dev%d.co" % k)
        open(one, "wb").write(blob[b0:b1])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + one,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        fn = None
        p = subprocess.Popen([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True)
        for line in p.stdout:
            m = head.match(line)
            if m:
                fn = m.group(1); counts.setdefault(fn, [0, 0])
            elif fn and "flat_load" in line:
                counts[fn][0] += 1
            elif fn and ("flat_store" in line or "flat_atomic" in line):
                counts[fn][1] += 1
        assert p.wait() == 0
    return counts


def test_step_kernels_have_no_flat_memory_access(flat_counts):
    step = {f: n for f, n in flat_counts.items() if re.search(r"general_arena_step|k_fused|k_quiet|k_step|k_reset|k_respawn|scan2", f)}
    assert any("general_arena_step" in f for f in step) and any("k_fused" in f for f in step), "kernels not found in the code object"
    # nothing in the step path writes through a flat address (the general engine writes LDS all the time: a generic LDS pointer shows up here)
    bad = {f: n for f, n in step.items() if n[1]}
    assert not bad, "flat_store / flat_atomic in the step path (LDS / HBM must be ds_* / global_*): %s" % sorted(bad.items())[:6]
    # kernels proper: no flat loads either.  general_arena_step (a real call) reads the descriptor through its generic `gs` argument --
    # read-only HBM, non-negative offsets from a global base -- and nothing else: well under 200 loads (with LDS behind a generic pointer
    # it had 600+ flat instructions)
    bad = {f: n for f, n in step.items() if n[0] and "general_arena_step" not in f}
    assert not bad, "flat_load in a step kernel: %s" % sorted(bad.items())[:6]
    assert all(n[0] < 200 for f, n in step.items()), sorted(step.items())[:4]
