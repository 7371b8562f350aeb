// The (kernel, device) table behind lds_attr.h.
#include <mutex>
#include <set>
#include <stdint.h>
#include <utility>

#include "lds_attr.h"

namespace nerfds {
bool lds_attr_first_use(const void* kernel, int device) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> seen;
  std::lock_guard<std::mutex> lock(mu);
  return seen.insert({kernel, device}).second;
}
}  // namespace nerfds

// include/nerfds.h: diagnostic hook for the CPU test of the guard's keying (no device call)
extern "C" int nerfds_debug_lds_attr_first_use(uint64_t kernel_key, int device) {
  return nerfds::lds_attr_first_use(reinterpret_cast<const void*>(static_cast<uintptr_t>(kernel_key)), device) ? 1 : 0;
}
