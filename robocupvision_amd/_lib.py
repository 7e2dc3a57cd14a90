"""ctypes binding of librcv.so (include/rcv.h).  No fallback: if the HIP library is missing or a
call fails, an exception is raised -- the package never computes the hot path any other way."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCV_LIBRARY: another build of the same library (A/B timing of two builds on one GPU box, scripts/ab.sh); product code never sets it
LIB_PATH = os.environ.get("RCV_LIBRARY") or os.path.join(_HERE, "librcv.so")

RCV_I_N, RCV_I_H, RCV_I_W, RCV_I_CIN, RCV_I_COUT, RCV_I_HO, RCV_I_WO, RCV_I_STRIDE, RCV_I_DIL, \
    RCV_I_INMODE, RCV_I_INMODE2, RCV_I_STATS, RCV_I_NPART, RCV_I_NSPLIT, RCV_I_COUNT, RCV_I_AUX0, RCV_I_AUX1 = range(17)
RCV_I__N = 20
(RCV_P_IN, RCV_P_IN_AUX, RCV_P_IN_C, RCV_P_W, RCV_P_BIAS, RCV_P_OUT, RCV_P_RESID, RCV_P_EPI_AUX, RCV_P_EPI_C,
 RCV_P_PART, RCV_P_IN2, RCV_P_IN2_AUX, RCV_P_IN2_C, RCV_P_X0, RCV_P_X1, RCV_P_X2, RCV_P_X3, RCV_P_X4, RCV_P_X5) = range(19)
RCV_P__N = 20

(OP_CONV, OP_TCONV, OP_WGRAD, OP_WGRAD_REDUCE, OP_PACK, OP_BN_FINALIZE, OP_BN_EVAL, OP_BN_BWD, OP_COMBINE, OP_CLS_FWD,
 OP_CLS_BWD, OP_CE_FWD, OP_CE_BWD, OP_POOL_FWD, OP_POOL_BWD, OP_ADAM_L1, OP_MEMSET, OP_CONV1X1, OP_ADD_SLICE,
 OP_MATERIALIZE, OP_BWD_STATS, OP_CONFUSION, OP_DICE_FWD, OP_DICE_BWD, OP_NHWC_TO_NCHW, OP_NCHW_TO_NHWC, OP_SGD, OP_NOP,
 OP_WGRAD_REDUCE_BATCH) = range(1, 30)

LOAD_PLAIN, LOAD_AFFINE, LOAD_GRAD_ENC, LOAD_GRAD_DEC, LOAD_NCHW, LOAD_AFFINE_RELU = range(6)
STATS_NONE, STATS_FWD, STATS_BWD_ENC, STATS_BWD_DEC = range(4)
F_BIAS, F_RELU, F_RESID, F_OUT_NCHW, F_FLIP, F_TRANSPOSED_SRC, F_ARGMAX, F_TRAINING, F_CONCAT = 1, 2, 4, 8, 16, 32, 64, 128, 256
F_FUSED_UP = 512
F_FUSED_CE = 1024
F_SIDE_STREAM = 1 << 16
F_MFMA_FP32 = 1 << 17      # contract on the fp32 matrix instructions only (rcv.h RCV_F_MFMA_FP32)


class RcvOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_uint32), ("i", C.c_int32 * RCV_I__N), ("f", C.c_float * 8),
                ("p", C.c_void_p * RCV_P__N)]


class RcvPackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("D0", C.c_int32), ("D1", C.c_int32),
                ("rows_from_d1", C.c_int32), ("flip", C.c_int32), ("rows_pad", C.c_int32), ("cols_pad", C.c_int32),
                ("merged", C.c_int32), ("reserved", C.c_int32), ("scale", C.c_void_p)]


class RcvReduceJob(C.Structure):      # struct rcv_reduce_job of include/rcv.h
    _fields_ = [("part", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("nsplit", C.c_int32), ("CB", C.c_int32),
                ("CA", C.c_int32), ("first_block", C.c_int32)]


EXPORTS = [
    "rcv_create", "rcv_destroy", "rcv_last_error", "rcv_version", "rcv_num_cus", "rcv_op_workspace", "rcv_run",
    "rcv_run_timed", "rcv_op_kernel_label", "rcv_run_ex", "rcv_join_side",
    "rcv_conv3x3", "rcv_convT3x3s2", "rcv_wgrad3x3", "rcv_bn_finalize", "rcv_bn_backward", "rcv_maxpool2x2_fwd",
    "rcv_softmax_ce_argmax_fwd", "rcv_softmax_ce_bwd", "rcv_adam_l1_step", "rcv_adam_l1_step_metrics", "rcv_confusion",
    "rcv_dice_fwd", "rcv_dice_bwd", "rcv_sgd_step", "rcv_create_planner", "rcv_adam_l1_step_pruned", "rcv_op_filter_layout",
]


class RcvError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()
_handles = {}


def load():
    """dlopen librcv.so (raises if it has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RcvError("librcv.so not found at %s -- build it with `make -C robocupvision_amd/csrc` "
                           "(there is no CPU fallback for the hot path)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.rcv_last_error.restype = C.c_char_p
        lib.rcv_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.rcv_create_planner.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.rcv_destroy.argtypes = [C.c_void_p]
        lib.rcv_num_cus.argtypes = [C.c_void_p]
        lib.rcv_op_workspace.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.POINTER(C.c_size_t)]
        lib.rcv_run.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_int, C.c_void_p]
        lib.rcv_run_ex.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_int, C.c_void_p, C.c_uint32]
        lib.rcv_join_side.argtypes = [C.c_void_p, C.c_void_p]
        lib.rcv_run_timed.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        lib.rcv_op_kernel_label.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_char_p, C.c_int]
        lib.rcv_op_filter_layout.argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_int]
        for name in ("rcv_conv3x3", "rcv_convT3x3s2", "rcv_wgrad3x3"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(RcvOp), C.c_void_p]
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rcv_last_error().decode("utf-8", "replace")
        raise RcvError("librcv %s failed (%d): %s" % (what, rc, msg))


def handle(device_index: int):
    """One library handle per device (created on first use)."""
    lib = load()
    h = _handles.get(device_index)
    if h is None:
        out = C.c_void_p()
        check(lib.rcv_create(device_index, C.byref(out)), "rcv_create")
        h = out
        _handles[device_index] = h
    return h


def planner_handle(num_cus: int = 256):
    """A handle without a device (rcv_create_planner): buffer layout / workspace queries only.  Used to lower a graph to its op
    lists on a host without a GPU (tests of the data-parallel bucket logic); every call that would enqueue work fails."""
    lib = load()
    h = _handles.get(("planner", num_cus))
    if h is None:
        out = C.c_void_p()
        check(lib.rcv_create_planner(num_cus, C.byref(out)), "rcv_create_planner")
        h = out
        _handles[("planner", num_cus)] = h
    return h


def join_side(h, stream_ptr: int):
    """Make the stream wait for everything the handle's side stream holds so far."""
    check(load().rcv_join_side(h, C.c_void_p(stream_ptr)), "rcv_join_side")


# RCV_MFMA_FP32=1: every contraction on the fp32 matrix instructions (the A/B switch for the split-bf16 kernels, rcv.h RCV_F_MFMA_FP32)
MATRIX_FLAGS = F_MFMA_FP32 if os.environ.get("RCV_MFMA_FP32") else 0


def make_op(kind: int, flags: int = 0, **kw) -> RcvOp:
    """Build a record; keyword names are the lower-cased slot names (n=, cin=, p_in=, f0=...)."""
    op = RcvOp()
    op.kind = kind
    op.flags = flags | (MATRIX_FLAGS if kind in (OP_CONV, OP_TCONV, OP_WGRAD) else 0)
    for k, v in kw.items():
        if k.startswith("p_"):
            idx = globals()["RCV_P_" + k[2:].upper()]
            op.p[idx] = v if v else None
        elif k.startswith("f") and k[1:].isdigit():
            op.f[int(k[1:])] = float(v)
        else:
            op.i[globals()["RCV_I_" + k.upper()]] = int(v)
    return op


def op_filter_layout(h, op: RcvOp, force: bool = False) -> int:
    """0: plain [9][Cin][Cout] filter; 2: the library wants the Winograd-transformed filter for this conv record."""
    return int(load().rcv_op_filter_layout(h, C.byref(op), 1 if force else 0))


def op_workspace(h, op: RcvOp) -> int:
    nbytes = C.c_size_t(0)
    check(load().rcv_op_workspace(h, C.byref(op), C.byref(nbytes)), "rcv_op_workspace")
    return int(nbytes.value)


class OpList:
    """A cached array of records executed by one rcv_run call."""

    def __init__(self, ops):
        self.n = len(ops)
        self.arr = (RcvOp * max(self.n, 1))(*ops)

    def run(self, h, stream_ptr: int):
        if self.n:
            check(load().rcv_run(h, self.arr, self.n, C.c_void_p(stream_ptr)), "rcv_run")

    def run_slice(self, h, stream_ptr: int, start: int, end: int, join: bool = True):
        """Ops [start, end); join=False leaves the side stream un-joined (see join_side)."""
        if end > start:
            first = C.cast(C.byref(self.arr, start * C.sizeof(RcvOp)), C.POINTER(RcvOp))
            check(load().rcv_run_ex(h, first, end - start, C.c_void_p(stream_ptr), 0 if join else 1), "rcv_run_ex")

    def run_timed(self, h, stream_ptr: int):
        """Profiling aid: per-op milliseconds (HIP events around every op; synchronises)."""
        ms = (C.c_float * max(self.n, 1))()
        if self.n:
            check(load().rcv_run_timed(h, self.arr, self.n, C.c_void_p(stream_ptr), ms), "rcv_run_timed")
        return [float(ms[k]) for k in range(self.n)]

    def labels(self, h):
        out = []
        buf = C.create_string_buffer(64)
        for k in range(self.n):
            check(load().rcv_op_kernel_label(h, C.byref(self.arr[k]), buf, 64), "rcv_op_kernel_label")
            out.append(buf.value.decode())
        return out
