"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/gtx.h declares
(no compute calls without a GPU), and fails loudly instead of falling back when there is no GPU."""
import ctypes
import os
import re

import pytest

import gtx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gtx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gtx_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("gtx_create", "gtx_destroy", "gtx_set_refs", "gtx_count", "gtx_count_device", "gtx_scan", "gtx_scan_device",
                 "gtx_last_error", "gtx_set_stream", "gtx_profile_read"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(gtx.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "libgtx.so does not export %s" % name


def test_python_binding_types_every_declared_symbol():
    assert sorted(gtx.ABI) == declared_symbols()
    gtx.load()


def test_version():
    assert gtx.load().gtx_version() >= 100


def test_scan_window_count_formula():
    lib = gtx.load()
    assert lib.gtx_scan_n_windows(10000, 1000, 1000) == 10
    assert lib.gtx_scan_n_windows(10000, 1000, 2000) == 9
    assert lib.gtx_scan_n_windows(700, 1000, 1000) == 0
    assert lib.gtx_scan_n_windows(2500, 25, 500) == 100 - 20 + 1


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = gtx.load()
    assert not lib.gtx_create(0)
    msg = lib.gtx_last_error(None).decode()
    assert "no usable HIP device" in msg and "no CPU path" in msg
    with pytest.raises(gtx.GtxError):
        gtx.Engine(0)


def test_product_does_not_touch_the_oracle():
    """Nothing under the product package may import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".c", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), "%s mentions the oracle" % os.path.join(dp, fn)
