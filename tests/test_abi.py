"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol that include/adaprompt_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os

import pytest

from adaprompt_amd import _lib, build


def test_library_builds_and_exports_header_symbols():
    path = build.build(verbose=False)
    assert os.path.exists(path)
    protos = _lib.parse_header()
    assert len(protos) >= 25
    lib = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/adaprompt_hip.h but not exported"
    lib.adap_abi_version.restype = ctypes.c_int
    assert lib.adap_abi_version() == _lib.ABI_VERSION          # bumped with every argument-list change


def test_bad_arguments_return_status_not_crash():
    """argument validation happens on the host before any launch, so it is checkable without a GPU."""
    lib = _lib.load()
    # Cin = 12 is not a multiple of 8 -> ADAP_ERR_ALIGN, message set
    rc = lib.adap_conv2d_nhwc(16, 1, 12, 16, 0, 0, 0, 0, 0, 16, 8, 0, 0, 1, 4, 4, 12, 4, 4, 8, 1, 1, 1, 0, 0, 1.0, 1, 0,
                              1, 0, 0, 0, 0, 0)
    assert rc == -2 and b"Cin" in lib.adap_last_error()
    with pytest.raises(_lib.HipError):
        _lib.call("adap_attention_fwd", 16, 40, 16, 40, 16, 40, 0, 0, 16, 40, 0, 1, 8, 16, 16, 5, 0.1, 0)   # d = 5


def test_product_package_does_not_import_oracle():
    import re
    pkg = os.path.dirname(os.path.abspath(build.__file__))
    repo = os.path.dirname(pkg)
    # the product package, the `ldm` alias package and the tuning tools: none may touch oracle/ (only tests/, smoke() and the
    # bench's cpu_baseline leg do)
    for root in (pkg, os.path.join(repo, "ldm"), os.path.join(repo, "tools")):
        for dp, _, files in os.walk(root):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
    bench_src = open(os.path.join(repo, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", bench_src, re.M)]
    start = bench_src.index("def cpu_baseline(")
    end = bench_src.index("\ndef ", start + 1)
    assert len(uses) == 1 and start < uses[0] < end, "bench.py may use the oracle inside cpu_baseline() only"


def test_ksplit_scale_switch_and_workspace_query_sized_for_the_largest_plan():
    """``adap_conv_ksplit_scale`` (the micro-batch lanes lower the split-K target while two streams keep the chip busy): returns
    the previous value, clamps to 0..100, a negative argument only reads -- and ``adap_conv2d_workspace_floats`` does not follow
    it (always the 100 % plan), so sizes a caller cached stay valid whatever the scale becomes.  Pure host code: no GPU needed."""
    lib = _lib.load()
    was = lib.adap_conv_ksplit_scale(-1)
    try:
        assert 0 <= was <= 100
        q = lambda: lib.adap_conv2d_workspace_floats(1, 1024, 1, 5120, 1280, 1, 1)      # FF2 at the 16 x 16 level: long K, few tiles
        lib.adap_conv_ksplit_scale(100)
        full = q()
        assert full > 0 and full % (1024 * 1280) == 0 and full // (1024 * 1280) >= 2
        assert lib.adap_conv_ksplit_scale(35) == 100 and q() == full
        assert lib.adap_conv_ksplit_scale(0) == 35 and q() == full
        assert lib.adap_conv_ksplit_scale(1000) == 0 and lib.adap_conv_ksplit_scale(-1) == 100
    finally:
        lib.adap_conv_ksplit_scale(was)
