// Fragment reads from LDS issued by hand, several fragments ahead of the MFMAs that use them (gfx950).
//
// Left to itself the compiler reads one fragment, waits for it and multiplies, with nothing in flight when the kernel
// has no registers to spare for a look-ahead of the compiler's own choosing.  A read the compiler cannot see is also one
// it cannot wait for: every use goes through frag_wait, which carries the fragment as an operand so that the MFMAs depend
// on it.  LDS operations complete in order, so the compiler's own reads and writes in between only make these waits
// stricter than they need to be, never too weak.
#pragma once
#include <type_traits>

#include "common.h"

namespace fcenc {

template <int B, int E, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// 16 bytes per lane from LDS byte address addr + IMM (IMM < 65536)
template <int IMM>
__device__ __forceinline__ void frag_read(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
// returns once at most N LDS operations issued after the read of f are still in flight
template <int N>
__device__ __forceinline__ void frag_wait(u32x4& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}

}  // namespace fcenc
