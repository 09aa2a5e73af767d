"""CPU-side checks of the C ABI: the library loads without a GPU and exports exactly what include/pt_hip.h
declares; calls that need a device fail with a status, not a crash."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pathtracer_amd import capi
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in pt_hip.h but not exported"
    assert sorted(capi.SIGNATURES) == names, "python binding table and header disagree"
    assert lib.pt_abi_version() == 7


def test_no_device_is_a_status_not_a_crash():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pathtracer_amd import capi
    lib = capi.load()
    ctx = C.c_void_p()
    st = lib.pt_create(0, C.byref(ctx))
    assert st == capi.PT_ERR_NO_DEVICE and not ctx.value
    assert len(lib.pt_last_error()) > 0
    with pytest.raises(capi.PtError):
        capi.Tracer(0)
    assert lib.pt_destroy(None) == capi.PT_OK


def test_product_never_touches_the_oracle():
    """Nothing under pathtracer_amd/ or include/ may reference oracle/ (it is test infrastructure)."""
    bad = []
    for base in ("pathtracer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    s = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"pt_oracle|oracle/|libpt_oracle", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
