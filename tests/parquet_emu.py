"""ctypes front of tests/emu/libparquet_emu.so: the product's Parquet reader (parquet_reader.hpp + parquet_device.hpp) run on the CPU,
thread by thread (test infrastructure; built on demand with g++)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "parquet_emu.cpp")
SO = os.path.join(HERE, "emu", "libparquet_emu.so")
CSRC = os.path.join(os.path.dirname(HERE), "polars_amd", "csrc")
_DEPS = [SRC] + [os.path.join(CSRC, h) for h in ("parquet_reader.hpp", "parquet_device.hpp", "parquet_snappy.hpp", "parquet_zstd.hpp", "parquet_zstd_index.hpp", "parquet_format.hpp", "host_codecs.hpp", "file_io.hpp")]
NP = {0: None, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.int64, 5: np.uint8, 6: np.uint16, 7: np.uint32, 8: np.uint64, 9: np.float32, 10: np.float64}

_lib = None


def build_if_stale(so, src, deps):
    """g++ -shared into a private temporary file, then an atomic rename: pytest-xdist workers may all find the library stale at once,
    and none of them may ever dlopen a half-written file."""
    if os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps):
        return
    tmp = f"{so}.{os.getpid()}.tmp"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", tmp, src, "-lpthread"], check=True)
    os.replace(tmp, so)


def lib():
    global _lib
    if _lib is None:
        build_if_stale(SO, SRC, _DEPS)
        l = C.CDLL(SO)
        l.pqemu_last_error.restype = C.c_char_p
        l.pqemu_read_column.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        l.pqemu_free.argtypes = [C.c_void_p]
        l.pqemu_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        l.pqemu_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        l.pqemu_category.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]
        l.pqemu_category.restype = C.c_int64
        l.pqemu_snappy.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        l.pqemu_snappy_host.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32]
        l.pqemu_host_codec.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32]
        l.pqemu_snappy_tag_selfcheck.argtypes = [C.c_uint32, C.c_uint64]
        l.pqemu_zstd.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        l.pqemu_snappy_tag_selfcheck.restype = C.c_int64
        _lib = l
    return _lib


class EmuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def read_column(path, row_groups, column, thread_order=0):
    """-> dict(values=np array (bool columns: unpacked), valid=np bool array or None, dtype, logical, null_count, categories, stats)"""
    l = lib()
    rg = (C.c_int * max(len(row_groups), 1))(*row_groups)
    h = C.c_void_p()
    rc = l.pqemu_read_column(path.encode(), rg, len(row_groups), column, thread_order, C.byref(h))
    if rc:
        raise EmuError(rc, l.pqemu_last_error().decode())
    try:
        dt, lg, n, nc, hv, ncat = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int64()
        st = (C.c_uint64 * 10)()
        l.pqemu_info(h, C.byref(dt), C.byref(lg), C.byref(n), C.byref(nc), C.byref(hv), C.byref(ncat), st)
        n = n.value
        nw = (n + 63) // 64
        if dt.value == 0:
            raw = np.zeros(nw, np.uint64)
        else:
            raw = np.zeros(n, NP[dt.value])
        vraw = np.zeros(nw, np.uint64)
        l.pqemu_copy(h, raw.ctypes.data_as(C.c_void_p), raw.nbytes, vraw.ctypes.data_as(C.c_void_p), vraw.nbytes if hv.value else 0)
        unpack = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)
        values = unpack(raw) if dt.value == 0 else raw
        valid = unpack(vraw) if hv.value else None
        cats = []
        buf = C.create_string_buffer(1 << 16)
        for i in range(ncat.value):
            ln = l.pqemu_category(h, i, buf, len(buf))
            cats.append(buf.raw[:ln])
        return {"values": values, "valid": valid, "dtype": dt.value, "logical": lg.value, "null_count": nc.value, "categories": cats,
                "stats": dict(zip(("file_bytes", "data_pages", "dict_pages", "snappy_streams", "snappy_bytes_out", "run_entries", "host_inflated_pages", "host_inflated_bytes", "zstd_streams", "zstd_blocks"), [int(x) for x in st])),
                "raw_validity_words": vraw if hv.value else None, "raw_value_words": raw if dt.value == 0 else None}
    finally:
        l.pqemu_free(h)


def snappy(data: bytes, n_out: int, thread_order=0):
    out = np.full(n_out + 64, 0x5A, np.uint8)
    rounds = C.c_uint32()
    err = lib().pqemu_snappy(data, len(data), out.ctypes.data_as(C.c_void_p), n_out, thread_order, C.byref(rounds))
    return err, out[:n_out].tobytes(), rounds.value, out[n_out:].tobytes()


def snappy_host(data: bytes, n_out: int):
    out = np.zeros(max(n_out, 1), np.uint8)
    rc = lib().pqemu_snappy_host(data, len(data), out.ctypes.data_as(C.c_void_p), n_out)
    return rc, out[:n_out].tobytes()


def host_codec(codec: str, data: bytes, n_out: int):
    """The product's host page decompressors ("zstd" / "lz4_raw") -> (rc, bytes, error text)."""
    out = np.zeros(max(n_out, 1), np.uint8)
    rc = lib().pqemu_host_codec({"zstd": 0, "lz4_raw": 1, "lz4_frame": 2, "gzip": 3}[codec], data, len(data), out.ctypes.data_as(C.c_void_p), n_out)
    return rc, out[:n_out].tobytes(), lib().pqemu_last_error().decode() if rc else ""


def zstd_device(data: bytes, n_out: int, thread_order=0):
    """The device zstd decoder's bodies (parquet_zstd.hpp) behind the host index pass -> (rc, bytes, counts, error text); rc 0 = ok, 64 = the
    kernels flagged the stream, -1 = the index pass rejected a header.  counts = blocks, compressed blocks, sequences, Huffman tables, FSE tables."""
    out = np.zeros(max(n_out, 1), np.uint8)
    counts = (C.c_uint32 * 8)()
    rc = lib().pqemu_zstd(data, len(data), out.ctypes.data_as(C.c_void_p), n_out, thread_order, counts)
    return rc, out[:n_out].tobytes(), list(counts)[:5], lib().pqemu_last_error().decode() if rc < 0 else ""
