// ipc_emu.cpp -- CPU access to the host half of the Arrow IPC scan (polars_amd/csrc/ipc_reader.hpp) for the tests: the bytes the product
// would upload for buffer `which` (0 validity, 1 values / offsets / views, 2.. data) of column `col` in record batch `batch`, after body
// decompression.  (TEST INFRASTRUCTURE: never linked into libpolars_amd.so.)
#include <cstring>

#include "../../polars_amd/csrc/ipc_reader.hpp"

using namespace plx::ipc;

static thread_local std::string t_err;

extern "C" {
const char* ipcemu_last_error() { return t_err.c_str(); }
// returns the buffer's (decompressed) length, or -1 on error; copies min(len, cap) bytes to out
int64_t ipcemu_buffer(const char* path, int batch, int col, int which, uint8_t* out, int64_t cap) {
  try {
    std::unique_ptr<File> f = open_file(path);
    const BatchMeta& bm = f->batches.at((size_t)batch);
    Slot s = slot_of(*f, bm, col);
    std::vector<uint8_t> v = read_buffer(*f, bm, f->body_off.at((size_t)batch), bm.buffers.at(s.buf + (size_t)which));
    const int64_t n = (int64_t)v.size() - 16;
    if (n > 0 && cap > 0) memcpy(out, v.data(), (size_t)(n < cap ? n : cap));
    return n;
  } catch (const std::exception& e) { t_err = e.what(); return -1; }
}
}
