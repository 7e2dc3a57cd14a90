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
    assert ctypes.sizeof(L.RcvPackJob) == 8 + 8 + 4 * 8 + 8      # ... + the per-output-channel scale pointer (inference BatchNorm folding)


def test_no_cpu_fallback():
    import torch
    import robocupvision_amd.model as M
    m = M.ROBO_UNet()
    with pytest.raises(L.RcvError):
        m(torch.zeros(1, 3, 16, 16))
    with pytest.raises(L.RcvError):
        M.CrossEntropyLoss2d()(torch.zeros(1, 5, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_planner_handle_plans_but_cannot_enqueue():
    """rcv_create_planner: a handle without a device answers layout queries (identically to a 256-CU MI355X handle) and rejects
    every call that would enqueue work -- there is no way to reach a kernel launch without a real device handle."""
    h = L.planner_handle(256)
    op = L.make_op(L.OP_CONV, L.F_BIAS | L.F_RELU, n=2, h=30, w=40, cin=128, cout=128, ho=30, wo=40, stride=1, dil=1,
                   inmode=L.LOAD_AFFINE, stats=L.STATS_FWD)
    nbytes = L.op_workspace(h, op)
    assert nbytes > 0 and op.i[L.RCV_I_NPART] > 0 and nbytes == op.i[L.RCV_I_NPART] * 2 * 128 * 4
    again = L.make_op(L.OP_CONV, L.F_BIAS | L.F_RELU, n=2, h=30, w=40, cin=128, cout=128, ho=30, wo=40, stride=1, dil=1,
                      inmode=L.LOAD_AFFINE, stats=L.STATS_FWD)
    assert L.op_workspace(h, again) == nbytes and again.i[L.RCV_I_NPART] == op.i[L.RCV_I_NPART]      # cached plan, same answer
    lst = L.OpList([op])
    assert lst.labels(h)[0].startswith("conv_dma")
    with pytest.raises(L.RcvError, match="planning-only"):
        lst.run(h, 0)
    with pytest.raises(L.RcvError, match="planning-only"):
        lst.run_timed(h, 0)
    with pytest.raises(L.RcvError):
        L.planner_handle(0)


def test_shipped_library_does_not_read_the_environment():
    """Experiment knobs are compiled in only by `make EXPERIMENTS=1`: the shipped librcv.so has no getenv on its launch path."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("nm not available")
    assert not any(tok.split("@")[0] == "getenv" for tok in out.stdout.split()), "librcv.so imports getenv: an experiment build was left in the tree"


def test_planner_accepts_every_layer_shape_of_the_networks():
    """Planner sweep (no GPU): every conv / transposed-conv / filter-gradient record the ROBO-UNet, U-Net and PB_FCN graphs produce at the
    BASELINE sizes and at ragged / tiny planes gets a plan (tile within the LDS limit, workspace size, label), for 256 CUs and for a small
    part (64 CUs).  A shape for which tile selection finds nothing would otherwise only show up as a failed launch on the GPU."""
    planes = [(32, 480, 640), (64, 120, 160), (32, 240, 320), (2, 48, 64), (1, 32, 48), (3, 37, 53), (1, 9, 11), (5, 130, 70)]
    chans = [(3, 8), (8, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 64), (24, 40), (4, 12)]
    labels = set()
    for cus in (256, 64):
        h = L.planner_handle(cus)
        for (n, H, W) in planes:
            for (ci, co) in chans:
                for s in (1, 2):
                    for d in ((1, 2) if s == 1 else (1,)):
                        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
                        if ci >= 4:
                            for mode in (L.LOAD_AFFINE, L.LOAD_GRAD_ENC):
                                op = L.make_op(L.OP_CONV, L.F_BIAS, n=n, h=H, w=W, cin=ci, cout=co, ho=Ho, wo=Wo, stride=s, dil=d,
                                               inmode=mode, stats=L.STATS_FWD)
                                assert L.op_workspace(h, op) >= 0
                                labels.add(L.OpList([op]).labels(h)[0].split("<")[0])
                            op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=n, h=H, w=W, cin=ci, ho=Ho, wo=Wo, cout=co, stride=s, dil=d,
                                           inmode=L.LOAD_AFFINE, inmode2=L.LOAD_GRAD_ENC)
                            assert L.op_workspace(h, op) > 0 and op.i[L.RCV_I_NSPLIT] >= 1
                            labels.add(L.OpList([op]).labels(h)[0].split("<")[0])
                        elif d == 1 or s == 1:
                            op = L.make_op(L.OP_WGRAD, L.F_BIAS, n=n, h=H, w=W, cin=ci, ho=Ho, wo=Wo, cout=co, stride=s, dil=d,
                                           inmode=L.LOAD_NCHW, inmode2=L.LOAD_GRAD_ENC)
                            assert L.op_workspace(h, op) > 0
                            labels.add(L.OpList([op]).labels(h)[0].split("<")[0])
                if ci >= 4 and co % 4 == 0:
                    op = L.make_op(L.OP_TCONV, L.F_BIAS, n=n, h=H, w=W, cin=ci, cout=co, ho=2 * H, wo=2 * W, stride=2, dil=1,
                                   inmode=L.LOAD_AFFINE, stats=L.STATS_FWD)
                    assert L.op_workspace(h, op) >= 0
    assert {"wgrad_mfma", "wgrad_first", "conv_dma", "convs_mfma"} <= labels, labels
