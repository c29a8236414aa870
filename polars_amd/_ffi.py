"""ctypes binding of libpolars_amd.so (the C ABI in include/polars_amd.h).

This is the stand-in for the Rust ``extern "C"`` shim a Polars maintainer would add to
polars-mem-engine (INTEGRATION.md shows that shim): it declares exactly the symbols of
the header and nothing else.  The product path has NO CPU fallback: if the HIP
extension is missing or no GPU is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

# hardware queues for the reader's per-column streams (core.cpp init_device has the measurement): set here as well because another library of the process
# (torch.distributed creating the RCCL communicator) may start the HIP runtime before plx_init does; a value the user has set wins
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpolars_amd.so")

# ---- enums (values mirror include/polars_amd.h) ---------------------------------------
BOOL, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(11)
EQ, NE, LT, LE, GT, GE = range(6)
ADD, SUB, MUL, TRUE_DIV, FLOOR_DIV, MOD = range(6)
AND, OR, XOR = range(3)
AGG_SUM, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_LEN, AGG_FIRST = range(7)
JOIN_INNER, JOIN_LEFT, JOIN_SEMI, JOIN_ANTI = range(4)
AE_COLUMN, AE_LITERAL, AE_BINARY, AE_CAST, AE_AGG, AE_LEN, AE_ALIAS, AE_NOT, AE_IS_NULL, AE_IS_NOT_NULL, AE_FILL_NULL = range(11)
(OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_PLUS, OP_MINUS, OP_MULTIPLY, OP_TRUE_DIVIDE,
 OP_FLOOR_DIVIDE, OP_MODULUS, OP_AND, OP_OR, OP_XOR) = range(15)
IR_SCAN, IR_FILTER, IR_SELECT, IR_HSTACK, IR_GROUPBY, IR_JOIN, IR_SORT, IR_SLICE = range(8)
PLAN_NO_FUSION = 1
PLAN_NO_DIRECT_JOIN = 2
PLAN_NO_PARTITION = 4
ERR_UNSUPPORTED = 3

DTYPE_WIDTH = {BOOL: 0, I8: 1, I16: 2, I32: 4, I64: 8, U8: 1, U16: 2, U32: 4, U64: 8, F32: 4, F64: 8}


class Scalar(C.Union):
    _fields_ = [("i", C.c_int64), ("u", C.c_uint64), ("f64", C.c_double), ("f32", C.c_float)]


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p), ("private_data", C.c_void_p),
]
ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class SeriesExport(C.Structure):
    _fields_ = [("field", C.POINTER(ArrowSchema)), ("arrays", C.POINTER(C.POINTER(ArrowArray))), ("len", C.c_size_t),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class AExpr(C.Structure):
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("lhs", C.c_int32), ("rhs", C.c_int32), ("dtype", C.c_int32),
                ("is_null", C.c_int32), ("lit", Scalar), ("name", C.c_char_p)]


class IR(C.Structure):
    _fields_ = [("kind", C.c_int32), ("input", C.c_int32), ("input_right", C.c_int32), ("predicate", C.c_int32),
                ("frame", C.c_uint64), ("exprs", C.POINTER(C.c_int32)), ("n_exprs", C.c_int32),
                ("keys", C.POINTER(C.c_int32)), ("n_keys", C.c_int32), ("keys_right", C.POINTER(C.c_int32)),
                ("n_keys_right", C.c_int32), ("how", C.c_int32), ("maintain_order", C.c_int32), ("suffix", C.c_char_p),
                ("sort_descending", C.POINTER(C.c_uint8)), ("sort_nulls_last", C.POINTER(C.c_uint8)), ("slice_offset", C.c_int64),
                ("slice_len", C.c_int64)]


class ProfileRecord(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("start_us", C.c_double), ("end_us", C.c_double), ("algo_bytes", C.c_uint64),
                ("rows", C.c_uint64)]


class PlxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[plx status {code}] {msg}")
        self.code = code
        self.msg = msg


class UnsupportedError(PlxError):
    """The node / dtype is outside the GPU hot path: run that subtree on the CPU engine
    (same contract as docs/source/user-guide/gpu-support.md)."""


_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

# every symbol declared in include/polars_amd.h: name -> (restype, argtypes)
SIGNATURES = {
    "plx_version": (C.c_uint32, []),
    "plx_last_error": (C.c_char_p, []),
    "plx_init": (C.c_int, [C.c_int]),
    "plx_shutdown": (C.c_int, []),
    "plx_set_stream": (C.c_int, [C.c_void_p]),
    "plx_synchronize": (C.c_int, []),
    "plx_set_cancel": (C.c_int, [C.c_int]),
    "plx_device_info": (C.c_int, [C.c_char_p, C.c_size_t, _i32p, _u64p]),
    "plx_memory_stats": (C.c_int, [_u64p, _u64p]),
    "plx_memory_trim": (C.c_int, []),
    "plx_memory_reserve": (C.c_int, [C.c_uint64]),
    "plx_column_from_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, _u64p]),
    "plx_column_from_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, _u64p]),
    "plx_column_placeholder": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, _u64p]),
    "plx_column_set_bounds": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64]),
    "plx_column_drop_statistics": (C.c_int, [C.c_uint64]),
    "plx_strview_dict_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, _u64p, _u64p]),
    "plx_strview_dict_encode_device": (C.c_int, [C.c_uint64, C.c_uint64, _u64p, _u64p]),
    "plx_strview_groupby": (C.c_int, [C.c_uint64, C.c_uint64, _u64p, _u64p, _u64p, _u64p, _u64p]),
    "plx_strview_stamp_nulls": (C.c_int, [C.c_uint64, C.c_uint64]),
    "plx_strdict_info": (C.c_int, [C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "plx_strdict_to_host": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p]),
    "plx_strdict_free": (C.c_int, [C.c_uint64]),
    "plx_datagen_id_views": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, _u64p]),
    "plx_datagen_long_id_views": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, _u64p, _u64p]),
    "plx_parquet_open": (C.c_int, [C.c_char_p, _u64p]),
    "plx_parquet_close": (C.c_int, [C.c_uint64]),
    "plx_parquet_shape": (C.c_int, [C.c_uint64, _i64p, _i32p, _i32p]),
    "plx_parquet_column_info": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_char_p), _i32p, _i32p, _i32p]),
    "plx_parquet_column_timezone": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_char_p)]),
    "plx_parquet_row_group_info": (C.c_int, [C.c_uint64, C.c_int32, _i64p, _i64p]),
    "plx_parquet_chunk_info": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, _i32p, C.POINTER(C.c_uint32), _i64p, _i64p, _i32p, C.POINTER(Scalar), C.POINTER(Scalar), _i64p]),
    "plx_parquet_read": (C.c_int, [C.c_uint64, _i32p, C.c_int32, _i32p, C.c_int32, _u64p]),
    "plx_parquet_categories": (C.c_int, [C.c_uint64, C.c_int32, _i64p, _i64p]),
    "plx_parquet_categories_to_host": (C.c_int, [C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "plx_parquet_column_strdict": (C.c_int, [C.c_uint64, C.c_int32, _u64p]),
    "plx_ipc_open": (C.c_int, [C.c_char_p, _u64p]),
    "plx_ipc_close": (C.c_int, [C.c_uint64]),
    "plx_ipc_shape": (C.c_int, [C.c_uint64, _i64p, _i32p, _i32p]),
    "plx_ipc_column_info": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_char_p), _i32p, _i32p, _i32p]),
    "plx_ipc_column_timezone": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_char_p)]),
    "plx_ipc_batch_info": (C.c_int, [C.c_uint64, C.c_int32, _i64p, _i64p, _i32p]),
    "plx_ipc_read": (C.c_int, [C.c_uint64, _i32p, C.c_int32, _i32p, C.c_int32, _u64p]),
    "plx_ipc_read_string_views": (C.c_int, [C.c_uint64, _i32p, C.c_int32, C.c_int32, _u64p, _u64p]),
    "plx_ipc_categories": (C.c_int, [C.c_uint64, C.c_int32, _i64p, _i64p]),
    "plx_ipc_categories_to_host": (C.c_int, [C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "plx_ipc_column_strdict": (C.c_int, [C.c_uint64, C.c_int32, _u64p]),
    "plx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "plx_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _u64p]),
    "plx_comm_info": (C.c_int, [C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "plx_comm_free": (C.c_int, [C.c_uint64]),
    "plx_exchange_by_key": (C.c_int, [C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, _u64p, _u64p, _u64p]),
    "plx_allgather_frame": (C.c_int, [C.c_uint64, C.c_uint64, _u64p]),
    "plx_column_import_arrow": (C.c_int, [C.POINTER(ArrowArray), C.POINTER(ArrowSchema), _u64p]),
    "plx_column_import_series": (C.c_int, [C.POINTER(SeriesExport), _u64p]),
    "plx_column_export_arrow": (C.c_int, [C.c_uint64, C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    "plx_column_export_series": (C.c_int, [C.c_uint64, C.c_char_p, C.POINTER(SeriesExport)]),
    "plx_column_to_host": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, _i32p]),
    "plx_column_copy_to_device": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p]),
    "plx_column_info": (C.c_int, [C.c_uint64, _i32p, _i64p, _i64p]),
    "plx_column_device_ptrs": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "plx_column_retain": (C.c_int, [C.c_uint64]),
    "plx_column_free": (C.c_int, [C.c_uint64]),
    "plx_cmp": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, _u64p]),
    "plx_cmp_scalar": (C.c_int, [C.c_int, C.c_uint64, Scalar, _u64p]),
    "plx_bitmap_binop": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, _u64p]),
    "plx_bitmap_not": (C.c_int, [C.c_uint64, _u64p]),
    "plx_arith": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, _u64p]),
    "plx_arith_scalar": (C.c_int, [C.c_int, C.c_uint64, Scalar, C.c_int, _u64p]),
    "plx_cast": (C.c_int, [C.c_uint64, C.c_int, _u64p]),
    "plx_filter": (C.c_int, [C.c_uint64, C.c_uint64, _u64p]),
    "plx_gather": (C.c_int, [C.c_uint64, C.c_uint64, _u64p]),
    "plx_reduce": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(Scalar), _i32p, _i32p]),
    "plx_groupby_agg": (C.c_int, [_u64p, C.c_int32, _u64p, _i32p, C.c_int32, C.c_int32, _u64p, _u64p]),
    "plx_join_indices": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, _u64p, _u64p]),
    "plx_datagen_lineitem_q1": (C.c_int, [C.c_int64, C.c_uint64, _u64p]),
    "plx_datagen_lineitem_q1_host": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plx_datagen_orders_lineitem": (C.c_int, [C.c_int64, C.c_uint64, _u64p, _u64p]),
    "plx_datagen_orders_lineitem_host": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64p]),
    "plx_datagen_uniform": (C.c_int, [C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_double, _u64p]),
    "plx_datagen_customer": (C.c_int, [C.c_int64, C.c_uint64, _u64p]),
    "plx_datagen_customer_host": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "plx_datagen_uniform_host": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_double, C.c_void_p]),
    "plx_datagen_zipf": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int64, _u64p]),
    "plx_datagen_zipf_host": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int64, C.c_void_p]),
    "plx_debug_program_json": (C.c_int, [C.POINTER(IR), C.c_int32, C.POINTER(AExpr), C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    "plx_sort_indices": (C.c_int, [_u64p, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int64, _u64p]),
    "plx_hash_partition": (C.c_int, [C.c_uint64, C.c_int32, C.c_uint64, _u64p, _i64p]),
    "plx_frame_new": (C.c_int, [C.POINTER(C.c_char_p), _u64p, C.c_int32, _u64p]),
    "plx_frame_free": (C.c_int, [C.c_uint64]),
    "plx_frame_concat": (C.c_int, [_u64p, C.c_int32, _u64p]),
    "plx_frame_shape": (C.c_int, [C.c_uint64, _i64p, _i32p]),
    "plx_frame_column": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_char_p), _u64p]),
    "plx_frame_dtypes": (C.c_int, [C.c_uint64, _i32p]),
    "plx_frame_to_host": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32p]),
    "plx_execute_plan": (C.c_int, [C.POINTER(IR), C.c_int32, C.POINTER(AExpr), C.c_int32, C.c_int32, C.c_uint32, _u64p]),
    "plx_describe_fusion": (C.c_int, [C.POINTER(IR), C.c_int32, C.POINTER(AExpr), C.c_int32, C.c_int32, _i32p, _i32p, C.c_char_p, C.c_size_t]),
    "plx_jit_selftest": (C.c_int, [C.POINTER(IR), C.c_int32, C.POINTER(AExpr), C.c_int32, C.c_int32]),
    "plx_jit_stats": (C.c_int, [_i32p, C.POINTER(C.c_double)]),
    "plx_jit_set_min_rows": (C.c_int, [C.c_int64]),
    "plx_last_plan_description": (C.c_char_p, []),
    "plx_profile_enable": (C.c_int, [C.c_int]),
    "plx_profile_fetch": (C.c_int, [C.POINTER(ProfileRecord), C.c_int32, _i32p]),
    "plx_profile_clear": (C.c_int, []),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP extension. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C polars_amd/csrc). "
                "There is no CPU fallback in polars_amd.")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = lib().plx_last_error().decode("utf-8", "replace")
        if status == ERR_UNSUPPORTED:
            raise UnsupportedError(status, msg)
        raise PlxError(status, msg)


_initialised = False


def init(device: Optional[int] = None) -> None:
    """Bind this process to one GPU (one process per GPU; LOCAL_RANK selects it by default)."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(lib().plx_init(device))
    _initialised = True


def ensure_init() -> None:
    if not _initialised:
        init()


_tls = threading.local()    # per calling thread, like plx_last_plan_description itself: the description of a collect() that was served by more than one
                            # library call (frame._string_key_group_by); reset by every collect()


def set_plan_note(note) -> None:
    _tls.plan_note = note


def last_plan() -> str:
    note = getattr(_tls, "plan_note", None)
    return note if note is not None else lib().plx_last_plan_description().decode()


def jit_set_min_rows(min_rows: int) -> None:
    """Inputs of at least ``min_rows`` rows get a run-time specialised kernel (hiprtc); negative disables the JIT."""
    check(lib().plx_jit_set_min_rows(int(min_rows)))


def jit_stats():
    n, ms = C.c_int32(), C.c_double()
    check(lib().plx_jit_stats(C.byref(n), C.byref(ms)))
    return n.value, ms.value
