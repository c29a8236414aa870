"""Test helper: calls libpolars_amd.so the way the reference's plugin loader does
(crates/polars-plan/src/plans/aexpr/function_expr/plugin.rs:70-137): SeriesExport inputs built from pyarrow arrays (Arrow C
Data Interface), ownership handed to the callee, result imported back into pyarrow."""
import ctypes as C
import pickle

import pyarrow as pa

from polars_amd import _ffi as F

RELEASE = C.CFUNCTYPE(None, C.POINTER(F.SeriesExport))
PLUGIN = C.CFUNCTYPE(None, C.POINTER(F.SeriesExport), C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(F.SeriesExport), C.c_void_p)
FIELD = C.CFUNCTYPE(None, C.POINTER(F.ArrowSchema), C.c_size_t, C.POINTER(F.ArrowSchema), C.c_char_p, C.c_size_t)


class Inputs:
    """n SeriesExports in one contiguous array (what `input.as_ptr()` is on the Rust side); counts the release calls."""

    def __init__(self, arrays, names=None):
        self.n = len(arrays)
        self.arr = (F.SeriesExport * self.n)()
        self.released = 0
        self._keep = []

        def on_release(p):
            s = p.contents
            for i in range(s.len):
                a = s.arrays[i].contents
                if a.release:
                    C.CFUNCTYPE(None, C.POINTER(F.ArrowArray))(a.release)(s.arrays[i])
            if s.field and s.field.contents.release:
                C.CFUNCTYPE(None, C.POINTER(F.ArrowSchema))(s.field.contents.release)(s.field)
            s.release = None
            self.released += 1
        self._cb = RELEASE(on_release)
        for i, a in enumerate(arrays):
            chunks = a.chunks if isinstance(a, pa.ChunkedArray) else [a]
            aa = [F.ArrowArray() for _ in chunks]
            sch = F.ArrowSchema()
            for j, ch in enumerate(chunks):
                tmp = F.ArrowSchema()
                ch._export_to_c(C.addressof(aa[j]), C.addressof(sch if j == 0 else tmp))
                if j:
                    C.CFUNCTYPE(None, C.POINTER(F.ArrowSchema))(tmp.release)(C.byref(tmp))
            if names:
                self._keep.append(names[i].encode())
                sch.name = self._keep[-1]
            ptrs = (C.POINTER(F.ArrowArray) * len(aa))(*[C.pointer(x) for x in aa])
            self._keep += [aa, sch, ptrs]
            self.arr[i].field = C.pointer(sch)
            self.arr[i].arrays = ptrs
            self.arr[i].len = len(aa)
            self.arr[i].release = C.cast(self._cb, C.c_void_p)
            self.arr[i].private_data = C.cast(C.pointer(sch), C.c_void_p)


def last_error() -> str:
    f = F.lib()._polars_plugin_get_last_error_message
    f.restype = C.c_char_p
    return (f() or b"").decode()


def call(name: str, arrays, kwargs=None, parallel=False, names=None):
    """-> (pyarrow array or None on failure, Inputs).  On failure last_error() holds the message."""
    lib = F.lib()
    fn = PLUGIN(("_polars_plugin_" + name, lib))
    inp = Inputs(arrays, names)
    out = F.SeriesExport()
    kw = pickle.dumps(kwargs, protocol=5) if kwargs else b""
    ctx = C.c_uint64(1 if parallel else 0)
    fn(inp.arr, inp.n, kw, len(kw), C.byref(out), C.cast(C.byref(ctx), C.c_void_p))
    if not out.private_data:
        return None, inp
    assert out.len == 1
    name_out = bytes(out.field.contents.name or b"")          # copied before the schema is consumed by the import below
    res = pa.Array._import_from_c(C.addressof(out.arrays[0].contents), C.addressof(out.field.contents))
    inp.out_name = name_out
    C.CFUNCTYPE(None, C.POINTER(F.SeriesExport))(out.release)(C.byref(out))
    return res, inp


def field(name: str, pa_type, kwargs=None, col_name="a"):
    """-> Arrow format string of the plugin's output field for one input of `pa_type`."""
    lib = F.lib()
    fn = FIELD(("_polars_plugin_field_" + name, lib))
    sch = F.ArrowSchema()
    pa.field(col_name, pa_type)._export_to_c(C.addressof(sch))
    out = F.ArrowSchema()
    kw = pickle.dumps(kwargs, protocol=5) if kwargs else b""
    fn(C.byref(sch), 1, C.byref(out), kw, len(kw))
    C.CFUNCTYPE(None, C.POINTER(F.ArrowSchema))(sch.release)(C.byref(sch))
    if not out.release:
        return None
    f = pa.Field._import_from_c(C.addressof(out))
    return f
