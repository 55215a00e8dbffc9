// Host-side helpers shared by the C-ABI entry points (error slot, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <feta_hip.h>

namespace feta {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Raises a kernel's dynamic-LDS cap above the 64 KB default.  The attribute is sticky, so it is set
// when the requested size grows, not on every launch (an eager launch loop is otherwise bound by this
// host call; inside a captured hipGraph it would not matter).  `seen` is a per-instantiation static.
template <class K>
inline void allow_dynamic_lds(K kern, size_t bytes, size_t& seen) {
  if (bytes > 64 * 1024 && bytes > seen) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
    seen = bytes;
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
