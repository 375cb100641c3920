// vm_buf.h — a device buffer whose physical pages arrive WHILE it is being filled.
//
// On this stack a large allocation costs ~27-30 ms per GB whatever the allocator (hipMalloc, hipMallocAsync with a retaining pool,
// hipMemCreate: the pages are cleared before they are handed out — profiles/r03_placement.md, r05_table_build.md: 160 GiB = 4.3-4.9 s,
// first call or not), and the per-edge tables are the largest thing a biased walk allocates (172 GB at config 3, 156 GB at config 5's
// stand-in): 5 s of a 9-16 s cold start spent waiting for cleared pages before the first table kernel could run.  The tables are
// written front to back (the work list is in row order, and so are the table offsets), so the buffer is ONE virtual range
// (hipMemAddressReserve: the kernels' addressing does not change) whose chunks of 4 GiB are created and mapped by a host thread while
// the build kernels fill what is already there; the build launches one segment of its work list per mapped chunk (edge_tables.hip).
// Anything the VMM calls refuse falls back to one hipMalloc.  SRW_EB_NO_VMM=1: always the plain allocation.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

namespace srw {

// set when a mapping call failed for a reason other than memory: the process then allocates its table buffers in one piece
inline std::atomic<bool> &vm_buf_broken() { static std::atomic<bool> b{false}; return b; }

template <typename T>
struct VmBuf {
  T *p = nullptr;
  size_t n = 0;                                  // elements
  size_t CHUNK = (size_t)4 << 30;                // bytes per mapped chunk (SRW_EB_VMM_CHUNK_MB: tests map small tables in many chunks)
  VmBuf() = default;
  VmBuf(const VmBuf &) = delete;
  VmBuf &operator=(const VmBuf &) = delete;
  VmBuf(VmBuf &&o) noexcept { take(o); }
  VmBuf &operator=(VmBuf &&o) noexcept { if (this != &o) { release(); take(o); } return *this; }
  ~VmBuf() { release(); }

  bool progressive() const { return vmm_; }
  size_t chunk_bytes() const { return CHUNK; }

  void release() {
    if (mapper_.joinable()) { stop_.store(true); mapper_.join(); }
    if (vmm_) {
      // chunk by chunk, as they were mapped: hipMemUnmap works per mapping — one call over a range of several mappings may undo
      // only the first (up to ~170 GB of HBM and the address range would stay pinned behind a (p, q) change or the NOMEM retry)
      const size_t mapped = mapped_.load();
      auto complain = [](const char *what, size_t off, hipError_t e) {
        fprintf(stderr, "[stellar_rw] VmBuf::release: %s at %zu: %s\n", what, off, hipGetErrorString(e));
        (void)hipGetLastError();
      };
      for (size_t off = 0; off < mapped; off += CHUNK) {
        const size_t sz = std::min(CHUNK, mapped - off);
        if (hipError_t e = hipMemUnmap((char *)p + off, sz); e != hipSuccess) complain("hipMemUnmap", off, e);
      }
      for (size_t i = 0; i < handles_.size(); ++i)
        if (hipError_t e = hipMemRelease(handles_[i]); e != hipSuccess) complain("hipMemRelease", i * CHUNK, e);
      handles_.clear();
      if (p) { if (hipError_t e = hipMemAddressFree((void *)p, va_bytes_); e != hipSuccess) complain("hipMemAddressFree", 0, e); }
    } else if (p) (void)hipFree(p);
    p = nullptr; n = 0; vmm_ = false; va_bytes_ = 0; mapped_.store(0); failed_.store(false); stop_.store(false); err_.clear();
  }

  // one allocation, complete on return (DevBuf::alloc)
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      throw Error(e == hipErrorOutOfMemory ? SRW_ERR_NOMEM : SRW_ERR_HIP, std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + " B): " + hipGetErrorString(e));
    }
    n = count; mapped_.store(count * sizeof(T));
  }

  // the virtual range now, the pages chunk by chunk from a host thread; wait_mapped() before touching a byte range
  void alloc_progressive(size_t count, int device) {
    release();
    if (count == 0) count = 1;
    const size_t bytes = count * sizeof(T);
    const char *no = getenv("SRW_EB_NO_VMM");
    if (const char *c = getenv("SRW_EB_VMM_CHUNK_MB"); c && atoi(c) >= 2) CHUNK = ((size_t)atoi(c) >> 1 << 1) << 20;     // (a multiple of 2 MiB)
    if (bytes < 2 * CHUNK || (no && *no == '1') || vm_buf_broken().load()) { alloc(count); return; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || CHUNK % gran) {
      (void)hipGetLastError(); alloc(count); return;
    }
    // every chunk is a whole CHUNK, the last one included (up to 4 GiB of slack): hipMemSetAccess on a second mapping of ANOTHER size inside
    // one reservation fails with "invalid argument" on this stack, erratically (scratch probe, round 5: 6 MiB / 1 GiB / 3.6 GB behind a
    // 4 GiB chunk failed, 4 GiB behind 4 GiB never did)
    // (the round-up is not in the planners' byte counts — their margins are: prepare_tables / prepare_shard_tables keep 8-40 GB
    //  beyond the tables' own bytes free, CHUNK is 4 GiB)
    const size_t total = (bytes + CHUNK - 1) / CHUNK * CHUNK;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, total, (size_t)2 << 20, nullptr, 0) != hipSuccess || !va) { (void)hipGetLastError(); alloc(count); return; }
    p = (T *)va; n = count; va_bytes_ = total; vmm_ = true;
    handles_.reserve((total + CHUNK - 1) / CHUNK);
    mapper_ = std::thread([this, prop, total, device] {
      (void)hipSetDevice(device);
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      for (size_t off = 0; off < total && !stop_.load(); off += CHUNK) {
        const size_t sz = std::min(CHUNK, total - off);
        hipMemGenericAllocationHandle_t h;
        const char *what = "hipMemCreate";
        hipError_t e = hipMemCreate(&h, sz, &prop, 0);
        if (const char *fa = getenv("SRW_EB_VMM_FAIL_AT"); fa && e == hipSuccess && (size_t)atoll(fa) == off / CHUNK) {     // tests: the fallback path
          (void)hipMemRelease(h); what = "hipMemCreate (simulated failure, SRW_EB_VMM_FAIL_AT)"; e = hipErrorInvalidValue;
        }
        if (e == hipSuccess) {
          what = "hipMemMap";
          e = hipMemMap((char *)p + off, sz, 0, h, 0);
          if (e == hipSuccess) { what = "hipMemSetAccess"; e = hipMemSetAccess((char *)p + off, sz, &acc, 1); }
          if (e != hipSuccess) (void)hipMemRelease(h); else handles_.push_back(h);
        }
        if (e != hipSuccess) {
          err_ = std::string("mapping the table buffer: ") + what + " of " + std::to_string(sz) + " B at " + std::to_string(off) + " of " + std::to_string(total) + " B: " + hipGetErrorString(e);
          oom_ = e == hipErrorOutOfMemory;
          if (!oom_) vm_buf_broken().store(true);          // (the caller retries: the next buffer is one hipMalloc)
          failed_.store(true); return;
        }
        mapped_.store(off + sz);
      }
    });
  }

  // blocks until bytes [0, upto) are backed; throws what the mapper met
  void wait_mapped(size_t upto_bytes) {
    if (!vmm_) return;
    if (upto_bytes > va_bytes_) upto_bytes = va_bytes_;
    while (mapped_.load() < upto_bytes) {
      if (failed_.load()) throw Error(oom_ ? SRW_ERR_NOMEM : SRW_ERR_HIP, err_);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  void wait_all() { wait_mapped(va_bytes_); if (mapper_.joinable()) mapper_.join(); if (failed_.load()) throw Error(oom_ ? SRW_ERR_NOMEM : SRW_ERR_HIP, err_); }

 private:
  void take(VmBuf &o) {                                      // (Graph is reset by move assignment: a buffer still being mapped is joined first)
    if (o.mapper_.joinable()) o.mapper_.join();
    p = o.p; n = o.n; CHUNK = o.CHUNK; vmm_ = o.vmm_; oom_ = o.oom_; va_bytes_ = o.va_bytes_; handles_ = std::move(o.handles_); err_ = std::move(o.err_);
    mapped_.store(o.mapped_.load()); failed_.store(o.failed_.load()); stop_.store(false);
    o.p = nullptr; o.n = 0; o.vmm_ = false; o.va_bytes_ = 0; o.handles_.clear(); o.mapped_.store(0); o.failed_.store(false);
  }
  bool vmm_ = false, oom_ = false;
  size_t va_bytes_ = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles_;
  std::thread mapper_;
  std::atomic<size_t> mapped_{0};
  std::atomic<bool> failed_{false}, stop_{false};
  std::string err_;
};

}  // namespace srw
