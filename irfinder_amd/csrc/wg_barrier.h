// wg_barrier.h — the workgroup barrier every kernel of this library uses.
//
// __syncthreads() with the wait for this wave's outstanding LDS operations spelled out.  In front of a barrier at a loop
// header the compiler was seen to emit none (ROCm 7.2, gfx950, count_kernel: `s_barrier` right behind the back edge's
// `ds_max_u32`, no `s_waitcnt lgkmcnt(0)`), and a read behind the barrier then missed an update issued in front of it —
// one run in a few dozen produced a wrong count.  The explicit wait costs nothing where the compiler's own is present.
#pragma once
#include <hip/hip_runtime.h>

namespace irf {
namespace dev {
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
}   // namespace dev
}   // namespace irf
