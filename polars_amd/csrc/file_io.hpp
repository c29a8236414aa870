// file_io.hpp -- positional reads of a local file for the scan sources (Parquet, Arrow IPC): host only, no HIP.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace plx {

struct IoError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Persistent helper threads for sliced reads.  A column chunk is a few megabytes and a slice thread that is CREATED for it costs about as much as the slice it copies
// (eight threads a chunk, twenty chunks a column: 4-8 ms of a 12 ms page walk were thread starts); the pool's threads are started once, the caller works through the
// queue itself while it waits (so a pool that is busy with other columns' slices never stalls a reader, and nested use cannot deadlock).  Never joined: like the
// library's other process-lifetime helpers it is left to the OS at exit.
class SlicePool {
 public:
  static SlicePool& get() { static SlicePool* p = new SlicePool(); return *p; }
  // runs every task (on pool threads and on the calling thread); returns when all are done; the first exception is rethrown
  void run(std::vector<std::function<void()>>& tasks) {
    if (tasks.empty()) return;
    if (tasks.size() == 1 || n_threads_ == 0) { for (auto& t : tasks) t(); return; }
    Group g;
    g.left = tasks.size();
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (auto& t : tasks) q_.push_back({&t, &g});
    }
    cv_.notify_all();
    for (;;) {            // help: take queued items (of any group) until this group is done
      Item it{nullptr, nullptr};
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (g.left == 0) break;
        if (!q_.empty()) { it = q_.front(); q_.pop_front(); }
        else { g.cv.wait(lk, [&] { return g.left == 0 || !q_.empty(); }); continue; }
      }
      exec(it);
    }
    if (g.err) std::rethrow_exception(g.err);
  }

 private:
  struct Group { size_t left = 0; std::exception_ptr err; std::condition_variable cv; };
  struct Item { std::function<void()>* fn; Group* g; };
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Item> q_;
  size_t n_threads_ = 0;
  void exec(const Item& it) {
    std::exception_ptr e;
    try { (*it.fn)(); } catch (...) { e = std::current_exception(); }
    std::lock_guard<std::mutex> lk(mu_);
    if (e && !it.g->err) it.g->err = e;
    if (--it.g->left == 0) it.g->cv.notify_all();
  }
  SlicePool() {
    const char* env = getenv("PLX_IO_THREADS");
    size_t n = env ? (size_t)std::max(0, atoi(env)) : std::min<size_t>(24, std::max(2u, std::thread::hardware_concurrency() / 4));
    n_threads_ = n;
    for (size_t i = 0; i < n; i++)
      std::thread([this] {
        for (;;) {
          Item it{nullptr, nullptr};
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !q_.empty(); });
            it = q_.front(); q_.pop_front();
          }
          exec(it);
        }
      }).detach();
  }
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
    const size_t kSlice = size_t(1) << 20;
    if (n < 2 * kSlice) { pread_exact(dst, n, off); return; }
    const size_t parts = std::min<size_t>(8, n / kSlice);
    const size_t per = (n / parts + 4095) & ~size_t(4095);
    std::vector<std::function<void()>> tasks;
    for (size_t t = 0; t < parts; t++) {
      const size_t b = std::min(n, t * per), e = std::min(n, (t + 1) * per);
      if (e > b) tasks.push_back([this, dst, off, b, e] { pread_exact((uint8_t*)dst + b, e - b, off + (int64_t)b); });
    }
    SlicePool::get().run(tasks);
  }
};

}  // namespace plx
