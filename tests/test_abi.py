"""The C-ABI library loads without a GPU and exports every symbol include/graphite_hip.h declares."""
import os
import re

from graphite_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "graphite_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    lib = _hip.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _hip.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_hip.SIGNATURES) == names


def test_xform_struct_layout_matches_header():
    assert _hip.XFORM_DTYPE.itemsize == 344
    assert _hip.XFORM_DTYPE.fields["crop_x"][1] == 72 and _hip.XFORM_DTYPE.fields["lut"][1] == 88


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.GraphiteHipError):
        _hip.ctx()
    from graphite_amd import transforms
    with pytest.raises(_hip.GraphiteHipError):
        transforms.add_noise(torch.zeros(3, 8, 8), torch.zeros(3, 4, 4), 1.0, torch.zeros(3, 4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "graphite_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                s = open(os.path.join(d, f)).read()
                assert "oracle_lib" not in s and "libgraphite_oracle" not in s and "graphite_oracle.h" not in s, f
