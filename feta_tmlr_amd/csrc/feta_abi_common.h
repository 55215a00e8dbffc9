// Host-side helpers shared by the C-ABI entry points (error slot, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <feta_hip.h>

namespace feta {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
}  // namespace feta

#define FETA_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      feta::set_error(__VA_ARGS__); \
      return FETA_E_ARG;            \
    }                               \
  } while (0)
