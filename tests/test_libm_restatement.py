"""agarcl_amd/csrc/agar_libm.inl (device sinf/cosf/atanf, restating glibc 2.35) vs the host libm.
Default: a strided sample of all float bit patterns + dense windows; AGAR_EXHAUSTIVE=1 checks all 2^32."""
import os
import subprocess

from conftest import ROOT


def _checker():
    out = os.path.join(ROOT, "tests", "_build", "libm_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-o", out, os.path.join(ROOT, "tests", "helpers", "libm_check.cpp"), "-lm"])
    return out


def _run(exe, start, count, stride):
    o = subprocess.check_output([exe, "%x" % start, str(count), str(stride)]).decode().split()
    return int(o[0]), int(o[1]), int(o[2]), o[3:]


def test_libm_restatement_sampled():
    exe = _checker()
    if os.environ.get("AGAR_EXHAUSTIVE") == "1":
        jobs = [(i << 28, 1 << 28, 1) for i in range(16)]
    else:
        jobs = [(0, (1 << 32) // 4099, 4099),            # strided over every exponent / sign
                (0x3f000000, 1 << 22, 1), (0x40490000, 1 << 20, 1), (0xc0000000, 1 << 21, 1),  # dense near 0.5, pi, -2
                (0x42f00000 - (1 << 18), 1 << 19, 1)]     # across the |x| = 120 reduction switch
    for start, count, stride in jobs:
        s, c, a, first = _run(exe, start, count, stride)
        assert (s, c, a) == (0, 0, 0), "mismatches sin=%d cos=%d atan=%d first=%s" % (s, c, a, first)
