"""The C-ABI library loads without a GPU and exports every symbol include/pymodem_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pymodem_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import pymodem_amd
    from pymodem_amd import _native
    path = pymodem_amd.library_path()
    assert os.path.exists(path), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert set(_native.EXPORTS) == set(names), set(_native.EXPORTS) ^ set(names)


def test_loading_does_not_need_or_touch_a_gpu():
    import pymodem_amd
    lib = pymodem_amd.lib()
    assert lib.pm_version() == 100
    assert lib.pm_device_count() >= 0
    if lib.pm_device_count() == 0:
        with pytest.raises(pymodem_amd.NativeError):          # no CPU fallback: the product path fails loudly
            pymodem_amd.Context(0)
        from pymodem_amd import chain_builder as cb
        import numpy as np
        m = cb.ModemConfigurator(48000, {"type": "fsk", "config": "9600", "options": {}})
        with pytest.raises(pymodem_amd.NativeError):
            m.demod(np.zeros(1000, dtype=np.int16))
