// Host-side helpers shared by the C-ABI entry points (error slot, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <feta_hip.h>

#include <atomic>

namespace feta {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Raises a kernel's dynamic-LDS cap above the 64 KB default.  The attribute is sticky PER DEVICE, so it is set
// when the requested size grows on the current device, not on every launch (an eager launch loop is otherwise
// bound by this host call; inside a captured hipGraph it would not matter).  `seen` is a per-instantiation
// static: one atomic high-water mark per device ordinal.  A refused attribute is reported through the error
// slot (note_attr_error) and surfaces with the launch error that follows it.
constexpr int kMaxDevices = 16;
struct LdsSeen {
  std::atomic<size_t> bytes[kMaxDevices];
  LdsSeen() {
    for (auto& b : bytes) b.store(0, std::memory_order_relaxed);
  }
};
void note_attr_error(const char* what);
void clear_attr_error();   // a note belongs to the launch that follows it, never to a later one

template <class K>
inline void allow_dynamic_lds(K kern, size_t bytes, LdsSeen& seen) {
  clear_attr_error();
  if (bytes <= 64 * 1024) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
  if (dev >= 0 && bytes <= seen.bytes[dev].load(std::memory_order_acquire)) return;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    note_attr_error(hipGetErrorString(e));
    return;
  }
  if (dev >= 0) {   // monotone maximum; two threads racing here both set an attribute that is large enough
    size_t cur = seen.bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !seen.bytes[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {
    }
  }
}
}  // namespace feta

#define FETA_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      feta::set_error(__VA_ARGS__); \
      return FETA_E_ARG;            \
    }                               \
  } while (0)
