// host_stage.hpp -- file bytes -> HBM through two page-locked staging buffers: buffer k + 1 is filled from the file while buffer k
// is on its way over PCIe (one DMA each, on the calling thread's stream).  One instance per host thread, kept for the thread's life:
// hipHostMalloc costs milliseconds, a column chunk takes about as long to upload.  No destructor on purpose (at process exit the
// HIP runtime may already be gone; the buffers are left to the OS like the library's bounce buffer in core.cpp).
#pragma once
#include <cstdlib>

#include "core.hpp"

namespace plx {

struct PinnedStage {
  // A staging buffer beyond kMaxPinnedBytes is ordinary pageable memory (the runtime stages its upload itself, and it is given back as soon as a smaller request
  // follows): 16 * n_rows bytes of string views at 1e9 rows would otherwise stay page-locked for the life of the process, twice per reader thread -- and a failed
  // hipHostMalloc must not fail a read that pageable memory can serve.
  static constexpr size_t kMaxPinnedBytes = size_t(512) << 20;
  struct Slot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; bool pinned = false; };
  Slot slot_[2];
  int next_ = 0;

  static void release(Slot& s) {
    if (!s.p) return;
    if (s.pinned) (void)hipHostFree(s.p); else free(s.p);
    s.p = nullptr; s.cap = 0; s.pinned = false;
  }
  // a host buffer of at least `bytes`, valid until the next-but-one call
  uint8_t* get(size_t bytes) {
    Slot& s = slot_[next_];
    next_ ^= 1;
    if (s.pending) { PLX_HIP(hipEventSynchronize(s.ev)); s.pending = false; }
    if (s.cap < bytes || (!s.pinned && s.p && s.cap > 4 * std::max(bytes, size_t(8) << 20))) {
      release(s);
      size_t cap = std::max(bytes + bytes / 8, size_t(8) << 20);      // headroom: the next column / chunk a little larger than this one does not re-allocate
      if (cap <= kMaxPinnedBytes && hipHostMalloc(&s.p, cap, hipHostMallocDefault) == hipSuccess) s.pinned = true;
      else {
        (void)hipGetLastError();
        s.p = malloc(cap);
        PLX_REQUIRE(s.p != nullptr, PLX_ERR_OOM, "host staging buffer: out of memory");
        s.pinned = false;
      }
      s.cap = cap;
    }
    if (!s.ev) PLX_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    return (uint8_t*)s.p;
  }
  // asynchronous copy of a buffer obtained from get(); the slot is reusable once the copy has run
  void upload(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    PLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()));
    for (Slot& s : slot_)          // any part of a staging buffer: the buffer is busy until this copy (and those queued before it) has run
      if (s.p && (const uint8_t*)src >= (const uint8_t*)s.p && (const uint8_t*)src < (const uint8_t*)s.p + s.cap) { PLX_HIP(hipEventRecord(s.ev, stream())); s.pending = true; }
  }
  static PinnedStage& for_this_thread() {
    static thread_local PinnedStage* st = new PinnedStage();    // never destroyed (see above)
    return *st;
  }
};

}  // namespace plx
