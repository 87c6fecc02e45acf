// walk_common.h — definitions shared by the jump kernels (walk_kernel_stream.h, walk_kernel_rot.h) and the host
// side of the engine (engine_capi.hip).
//
// Replaces the data layout part of comp_kangaroos / ComputeKangaroos (math/gpu/engine.cu:16-19,
// math/gpu/compute.h:5-65, math/gpu/kernel.h:175-254).
//
// Layout in HBM (engine-private, "SoA over kangaroos"): three arrays X, Y, D of 4 planes each;
// plane w of kangaroo p is at  base[w * cap + p]  (uint64 limb w, p = kangaroo index in host
// order, cap = padded kangaroo count).  Thread T owns kangaroos p = T + k * threads, k < K, so
// every 8-byte plane load/store is lane-contiguous.
// Distinguished points are appended by one atomicAdd per wavefront (ballot + prefix rank) into
// a host-visible pinned buffer.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fe29.h"
#include "modinv30.h"

namespace kg {

constexpr int NB_JUMP = 64;        // constants.h:12
constexpr int WALK_BLOCK = 256;    // 4 wavefronts per workgroup
constexpr int JT_STRIDE = 19;      // LDS row: 9 limbs jPx, 9 limbs jPy, 1 pad (odd stride spreads banks)

// device-format jump table built by the host in kg_set_params
struct JumpTableDev {
  u32 pxy[NB_JUMP][18];  // jPx limbs then jPy limbs (canonical, 29-bit)
  u32 dist[NB_JUMP][8];  // jD as 8 little-endian 32-bit words
};

struct DpItemDev {  // 80 bytes, written to pinned host memory
  u32 x[8];
  u32 d[8];
  u64 kidx;
  u64 pad;
};

struct WalkArgs {
  u64 *X, *Y, *D;            // 4 planes each
  u64 cap;                   // plane stride (padded kangaroo count)
  u64 n;                     // real kangaroo count (of this launch segment: positions p >= n are padding)
  u64 p_base;                // position of this segment's first kangaroo in the whole herd: kIdx = p_base + p
  u32 threads;               // total threads (grid * block)
  u32 nb_run;                // steps per launch (NB_RUN)
  u32 dp_mask_hi, dp_mask_lo;
  const JumpTableDev *jt;
  DpItemDev *out;            // host-visible pinned
  u32 *out_count;            // device counter
  u32 max_found;
};


// compile-time loops: "#pragma unroll" only *suggests* (the unroller's size threshold refuses a body of
// ~2500 instructions and the per-kangaroo state would then be indexed dynamically, i.e. live in scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int I, typename F>
__device__ __forceinline__ void static_for_down(F &&f) {
  if constexpr (I >= 0) {
    f(std::integral_constant<int, I>{});
    static_for_down<I - 1>(f);
  }
}

template <int DW>
__device__ __forceinline__ void delta_add(u32 (&d)[DW], const u32 *jd) {
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < DW; ++i) {
    c += (u64)d[i] + jd[i];
    d[i] = (u32)c;
    c >>= 32;
  }
}

}  // namespace kg
