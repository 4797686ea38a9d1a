"""The C-ABI library loads and exports every symbol include/agarcl_batch.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "agarcl_batch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(agarcl_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from agarcl_amd import _capi
    assert declared_symbols() == sorted(n for n, _, _ in _capi.SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    import ctypes
    from agarcl_amd import _capi, build
    build.build()
    lib = ctypes.CDLL(_capi.HIP_SO)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    _capi.bind(lib)
    assert lib.agarcl_last_error() is not None


def test_no_gpu_fails_loudly():
    """Without a HIP device agarcl_create must return an error (no CPU fallback)."""
    import torch
    from agarcl_amd import _capi
    if torch.cuda.is_available():
        return
    try:
        _capi.BatchedEngine(1)
    except _capi.AgarclError as e:
        assert e.code == -4
    else:
        raise AssertionError("engine creation succeeded without a GPU")


def test_product_never_imports_oracle():
    """agarcl_amd/ must not reference oracle/ or the test-only emulation build."""
    pkg = os.path.join(ROOT, "agarcl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inl", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libagarcl_emu" not in txt or f == "agar_engine.hip", f
                assert "libagar_oracle" not in txt and "libagar_ref" not in txt, f


def test_vec_header_and_binding_agree():
    """include/agarcl_vec.h (one host call per vector step) is implemented by the same library: header, binding and exports agree"""
    import ctypes
    from agarcl_amd import _capi, build
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "agarcl_vec.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(agarcl_[a-z_0-9]+)\s*\(", src)))
    assert names == sorted(n for n, _, _ in _capi.VEC_SYMBOLS) == ["agarcl_vec_reset", "agarcl_vec_step"]
    build.build()
    lib = ctypes.CDLL(_capi.HIP_SO)
    for name in names:
        assert hasattr(lib, name), name
    # the structures the binding passes are the header's: field for field
    fields = lambda body: [f.split()[-1].lstrip("*").split("[")[0] for f in re.findall(r"([^;{}]+);", body)]
    for cname, cls in (("agarcl_vec_spec", _capi.VecSpec), ("agarcl_vec_buffers", _capi.VecBuffers)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        assert fields(body) == [f[0] for f in cls._fields_], cname
