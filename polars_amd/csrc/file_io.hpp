// file_io.hpp -- positional reads of a local file for the scan sources (Parquet, Arrow IPC): host only, no HIP.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace plx {

struct IoError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct FileReader {
  std::string path;
  int fd = -1;
  int64_t size = 0;
  FileReader() = default;
  FileReader(const FileReader&) = delete;
  FileReader& operator=(const FileReader&) = delete;
  ~FileReader() { if (fd >= 0) ::close(fd); }

  void open(const std::string& p) {
    path = p;
    fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) throw IoError("cannot open " + p);
    struct stat st;
    if (::fstat(fd, &st) != 0) throw IoError("cannot stat " + p);
    size = st.st_size;
  }
  void pread_exact(void* dst, size_t n, int64_t off) const {
    uint8_t* p = (uint8_t*)dst;
    while (n) {
      ssize_t got = ::pread(fd, p, n, off);
      if (got <= 0) throw IoError("short read from " + path);
      p += got; off += got; n -= (size_t)got;
    }
  }
  // A buffer out of the page cache is a memcpy: one thread moves ~10 GB/s, PCIe takes ~56 GB/s.  Large reads are cut into
  // slices read concurrently (positional reads on one descriptor are independent).
  void pread_sliced(void* dst, size_t n, int64_t off) const {
    const size_t kSlice = size_t(2) << 20;
    size_t threads = std::min<size_t>(8, n / kSlice);
    if (threads < 2) { pread_exact(dst, n, off); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(threads);
    const size_t per = (n / threads + 4095) & ~size_t(4095);
    for (size_t t = 0; t < threads; t++) {
      const size_t b = std::min(n, t * per), e = std::min(n, (t + 1) * per);
      pool.emplace_back([this, dst, off, b, e, t, &errs] {
        try { if (e > b) pread_exact((uint8_t*)dst + b, e - b, off + (int64_t)b); } catch (...) { errs[t] = std::current_exception(); }
      });
    }
    for (std::thread& th : pool) th.join();
    for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
  }
};

}  // namespace plx
