"""CPU: the C-ABI library loads and exports every symbol include/rcv.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from robocupvision_amd import _lib as L


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "rcv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcv_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_header_symbols():
    assert os.path.exists(L.LIB_PATH), "librcv.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "include/rcv.h declares %s but librcv.so does not export it" % n
    assert set(L.EXPORTS) <= set(names)


def test_version_and_error_text():
    lib = L.load()
    assert lib.rcv_version() == 100
    assert isinstance(lib.rcv_last_error(), bytes)


def test_struct_layout_matches_header():
    # rcv_op: int32 kind, uint32 flags, int32 i[20], float f[8], void* p[20]
    assert ctypes.sizeof(L.RcvOp) == 4 + 4 + 4 * 20 + 4 * 8 + 8 * 20
    assert ctypes.sizeof(L.RcvPackJob) == 8 + 8 + 4 * 8


def test_no_cpu_fallback():
    import torch
    import robocupvision_amd.model as M
    m = M.ROBO_UNet()
    with pytest.raises(L.RcvError):
        m(torch.zeros(1, 3, 16, 16))
    with pytest.raises(L.RcvError):
        M.CrossEntropyLoss2d()(torch.zeros(1, 5, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))
