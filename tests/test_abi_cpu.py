"""CPU-only checks of the drop-in boundary: libpolars_amd.so loads, exports every symbol
include/polars_amd.h declares (and the ctypes table binds exactly that set), reports the plugin
ABI version of the reference (polars-ffi/src/lib.rs:12-13 -> (0, 1)), and fails LOUDLY -- status
code + last-error string, no crash, no CPU fallback -- when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "polars_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(plx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from polars_amd import _ffi
    lib = _ffi.lib()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/polars_amd.h but not exported"
    assert sorted(_ffi.SIGNATURES) == syms, "ctypes binding and header disagree"


def test_version_matches_reference_plugin_abi():
    from polars_amd import _ffi
    v = _ffi.lib().plx_version()
    assert (v >> 16, v & 0xFFFF) == (0, 1)


def test_no_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from polars_amd import _ffi
    lib = _ffi.lib()
    rc = lib.plx_init(0)
    assert rc == 2                                   # PLX_ERR_HIP
    assert b"no HIP device" in lib.plx_last_error()
    import numpy as np
    h = C.c_uint64()
    a = np.arange(4, dtype=np.int64)
    rc = lib.plx_column_from_host(_ffi.I64, a.ctypes.data_as(C.c_void_p), None, 0, 4, C.byref(h))
    assert rc == 2 and b"plx_init" in lib.plx_last_error()
    import polars_amd as pl
    with pytest.raises(pl.PlxError):
        pl.Series("a", a)                            # the python mirror has no fallback either


def test_placeholder_columns_reject_compute():
    from polars_amd import _ffi as F
    lib = F.lib()
    h = C.c_uint64()
    assert lib.plx_column_placeholder(F.I64, 100, 0, 0, 0, 0, C.byref(h)) == 0
    dt, n, nulls = C.c_int32(), C.c_int64(), C.c_int64()
    assert lib.plx_column_info(h.value, C.byref(dt), C.byref(n), C.byref(nulls)) == 0
    assert (dt.value, n.value, nulls.value) == (F.I64, 100, 0)
    out = C.c_uint64()
    s = F.Scalar(); s.i = 1
    assert lib.plx_cmp_scalar(F.GT, h.value, s, C.byref(out)) != 0
    assert lib.plx_column_free(h.value) == 0
    assert lib.plx_column_free(h.value) == 1 and b"invalid column handle" in lib.plx_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under polars_amd/ may reference it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "polars_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"\boracle\b|pyoracle|plx_oracle|orc_", txt) and not f.endswith("dist.py"):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    txt = open(os.path.join(ROOT, "polars_amd", "dist.py")).read()
    assert "import oracle" not in txt and "pyoracle" not in txt


def test_header_is_plain_c_and_links(tmp_path):
    """include/polars_amd.h must be consumable by a C compiler (the boundary a cgo / JNI / Rust-FFI binding sees): a C11 program
    using the kernel-level and plan-level entry points compiles with -Wall -Werror, links against the built library and, on a box
    without a GPU, gets a clean error code + message instead of a crash."""
    import subprocess
    src = tmp_path / "abi_example.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "polars_amd.h"

int main(void) {
  if (plx_version() != ((PLX_ABI_MAJOR << 16) | PLX_ABI_MINOR)) return 10;
  int rc = plx_init(0);
  if (rc != PLX_OK) {                       /* no GPU here: a status code and a message, nothing else */
    const char* msg = plx_last_error();
    if (!msg || !strlen(msg)) return 11;
    printf("init failed as expected: %d\n", rc);
    /* the rest must fail the same way, not crash */
    int64_t v[4] = {1, 2, 3, 4};
    plx_column col = 0;
    if (plx_column_from_host(PLX_I64, v, NULL, 0, 4, &col) == PLX_OK) return 12;
    return 0;
  }
  /* with a GPU: filter(a > 2).select(sum(a)) through the kernel-level entry points */
  int64_t v[4] = {1, 2, 3, 4};
  plx_column a = 0, mask = 0, kept = 0;
  plx_scalar two; two.i = 2;
  plx_scalar out; plx_dtype out_dtype = PLX_I64; int32_t out_valid = 0;
  if (plx_column_from_host(PLX_I64, v, NULL, 0, 4, &a)) return 20;
  if (plx_cmp_scalar(PLX_GT, a, two, &mask)) return 21;
  if (plx_filter(a, mask, &kept)) return 22;
  if (plx_reduce(PLX_AGG_SUM, kept, &out, &out_dtype, &out_valid)) return 23;
  printf("sum = %lld\n", (long long)out.i);
  plx_column_free(a); plx_column_free(mask); plx_column_free(kept);
  return out.i == 7 ? 0 : 24;
}
''')
    exe = tmp_path / "abi_example"
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "polars_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lpolars_amd", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)


def test_reference_plugin_entry_points_are_exported_and_fail_loudly_without_a_gpu():
    """_polars_plugin_* are what an unmodified Polars resolves after dlopen (plugin.rs:23-137): version (0, 1), every declared
    function + its field function exported, output fields correct without any GPU work, and -- on a box without a GPU -- a call
    releases its inputs (the callee owns them, plugin.rs:122-125), leaves out.private_data NULL and sets the thread-local message."""
    import pyarrow as pa
    import torch

    from polars_amd import _ffi
    from tests import plugin_abi as P
    lib = _ffi.lib()
    src = open(os.path.join(ROOT, "include", "polars_amd.h")).read()
    names = re.findall(r"PLX_DECLARE_PLUGIN\((plx_[a-z]+)\)", src)
    assert len(names) == 19
    for n in names:
        assert hasattr(lib, "_polars_plugin_" + n) and hasattr(lib, "_polars_plugin_field_" + n), n
    lib._polars_plugin_get_version.restype = C.c_uint32
    v = lib._polars_plugin_get_version()
    assert (v >> 16, v & 0xFFFF) == (0, 1)
    # output fields (no device work)
    assert P.field("plx_gt", pa.int64()).type == pa.bool_()
    assert P.field("plx_cmp", pa.float64(), {"op": "le"}).type == pa.bool_()
    assert P.field("plx_arith", pa.int32(), {"op": "truediv"}).type == pa.float64()
    assert P.field("plx_arith", pa.int32(), {"op": "add"}).type == pa.int32()
    assert P.field("plx_truediv", pa.float32()).type == pa.float32()
    assert P.field("plx_sum", pa.int16()).type == pa.int64() and P.field("plx_sum", pa.uint32()).type == pa.uint32()
    assert P.field("plx_mean", pa.int64()).type == pa.float64() and P.field("plx_mean", pa.float32()).type == pa.float32()
    f = P.field("plx_filter", pa.timestamp("us"), col_name="ts")
    assert f.type == pa.timestamp("us") and f.name == "ts"
    assert P.field("plx_sum", pa.large_string()) is None and "unsupported Arrow format" in P.last_error()
    if torch.cuda.is_available():
        return
    res, inp = P.call("plx_gt", [pa.array([1, 2, 3], pa.int64()), pa.array([2], pa.int64())])
    assert res is None and inp.released == 2
    assert "no HIP device" in P.last_error() or "plx_init" in P.last_error()
    res, inp = P.call("plx_sum", [pa.array([1.0, 2.0]), pa.array([1.0])])       # wrong arity: still releases what it was given
    assert res is None and inp.released == 2


def test_string_key_group_by_entry_checks_its_arguments_before_it_needs_a_device(tmp_path):
    """plx_strview_groupby / plx_ipc_read_string_views: what can be refused from the handles alone is refused with the right status and message; only
    then does the call need the GPU (and says so)."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.ipc as ipc
    from polars_amd import _ffi as F
    lib = F.lib()
    ERR_INVALID, ERR_SHAPE = 1, 5                     # include/polars_amd.h

    def ph(dtype, n):
        h = C.c_uint64()
        assert lib.plx_column_placeholder(dtype, n, 0, 0, 0, 0, C.byref(h)) == 0
        return h.value
    outs = [C.c_uint64() for _ in range(5)]
    refs = [C.byref(o) for o in outs]
    views, odd, v64, v32, short = ph(F.U64, 200), ph(F.U64, 201), ph(F.F64, 100), ph(F.I32, 100), ph(F.F64, 99)
    assert lib.plx_strview_groupby(views, v64, None, *refs[1:]) == ERR_INVALID and b"null pointer" in lib.plx_last_error()
    assert lib.plx_strview_groupby(odd, v64, *refs) == ERR_INVALID and b"2 n words" in lib.plx_last_error()
    assert lib.plx_strview_groupby(v64, v64, *refs) == ERR_INVALID                                   # views must be UInt64
    assert lib.plx_strview_groupby(views, short, *refs) == ERR_SHAPE and b"differ in length" in lib.plx_last_error()
    assert lib.plx_strview_groupby(views, v32, *refs) == F.ERR_UNSUPPORTED and b"Float64 / Int64" in lib.plx_last_error()
    st = lib.plx_strview_groupby(views, v64, *refs)                                                     # well-formed: now it needs the device
    assert st != 0 and st != F.ERR_UNSUPPORTED and (b"HIP" in lib.plx_last_error() or b"placeholder" in lib.plx_last_error()), lib.plx_last_error()
    # nulls of a raw view column: plx_strview_stamp_nulls takes the views and a Boolean column of the same length
    vb, vb_short = ph(F.BOOL, 100), ph(F.BOOL, 99)
    assert lib.plx_strview_stamp_nulls(odd, vb) == ERR_INVALID and b"2 n words" in lib.plx_last_error()
    assert lib.plx_strview_stamp_nulls(views, v64) == ERR_INVALID and b"Boolean" in lib.plx_last_error()
    assert lib.plx_strview_stamp_nulls(views, vb_short) == ERR_SHAPE and b"differ in length" in lib.plx_last_error()
    assert lib.plx_strview_stamp_nulls(views, vb) == ERR_INVALID and b"placeholder" in lib.plx_last_error()
    # the IPC side: a Utf8 column, a dictionary-encoded one, a numeric one
    path = str(tmp_path / "t.arrow")
    t = pa.table({"s": pa.array(["a", "bb", "ccc"], pa.string()), "d": pa.array(["x", "y", "x"]).dictionary_encode(), "i": pa.array([1, 2, 3], pa.int64())})
    with ipc.new_file(path, t.schema) as w:
        w.write_table(t)
    fh = C.c_uint64()
    assert lib.plx_ipc_open(path.encode(), C.byref(fh)) == 0
    b0 = (C.c_int32 * 1)(0)
    o1, o2 = C.c_uint64(), C.c_uint64()
    assert lib.plx_ipc_read_string_views(fh.value, b0, 1, 7, C.byref(o1), C.byref(o2)) == ERR_INVALID and b"column index" in lib.plx_last_error()
    assert lib.plx_ipc_read_string_views(fh.value, b0, 1, 1, C.byref(o1), C.byref(o2)) == F.ERR_UNSUPPORTED and b"dictionary-encoded" in lib.plx_last_error()
    assert lib.plx_ipc_read_string_views(fh.value, b0, 1, 2, C.byref(o1), C.byref(o2)) == F.ERR_UNSUPPORTED
    assert lib.plx_ipc_read_string_views(fh.value, (C.c_int32 * 1)(5), 1, 0, C.byref(o1), C.byref(o2)) == ERR_INVALID and b"record batch index" in lib.plx_last_error()
    assert lib.plx_ipc_read_string_views(fh.value, b0, 1, 0, None, C.byref(o2)) == ERR_INVALID
    st = lib.plx_ipc_read_string_views(fh.value, b0, 1, 0, C.byref(o1), C.byref(o2))                   # well-formed: needs the device
    assert st != 0 and st != F.ERR_UNSUPPORTED and b"HIP" in lib.plx_last_error(), lib.plx_last_error()
    assert lib.plx_ipc_close(fh.value) == 0
