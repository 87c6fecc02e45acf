// walk_kernel.h — the kangaroo jump kernel for gfx950 (hand-written HIP, wave64).
//
// Replaces comp_kangaroos / ComputeKangaroos (math/gpu/engine.cu:16-19, math/gpu/compute.h:5-65).
//
// Layout in HBM (engine-private, "SoA over kangaroos"): three arrays X, Y, D of 4 planes each;
// plane w of kangaroo p is at  base[w * cap + p]  (uint64 limb w, p = kangaroo index in host
// order, cap = padded kangaroo count).  Thread T owns kangaroos p = T + k * threads, k < K, so
// every 8-byte plane load/store is lane-contiguous.
//
// Per thread: K kangaroos stay in registers (x, y as 9x29-bit limbs, plus a DW-word distance
// delta) for all nb_run steps.  Per step:
//   forward : dx_k = x_k - jPx[x_k & 63];  q_k = q_{k-1} * dx_k   (prefix products -> LDS)
//   invert  : one constant-time safegcd inverse per lane (64 different inverses per wave pass)
//   backward: dx_k^-1 = u * q_{k-1};  u *= dx_k;  affine add;  delta_k += jD;  DP test
// = 5 multiplications + 1 squaring per jump, the reference's count.
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
  u64 n;                     // real kangaroo count
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

// W = minimum waves per SIMD the register allocation must allow (1: up to 512 VGPR+AGPR, 2: 256)
template <int K, int DW, int W>
__global__ void __launch_bounds__(WALK_BLOCK, W) kg_walk_kernel(WalkArgs a) {
  extern __shared__ u32 lds[];
  u32 *jt = lds;                                  // [64][JT_STRIDE]
  u32 *jdist = lds + NB_JUMP * JT_STRIDE;         // [64][DW]
  u32 *pref = jdist + NB_JUMP * DW;               // [K-1][9][WALK_BLOCK]

  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < NB_JUMP * 18; i += WALK_BLOCK) {
    u32 row = i / 18, col = i % 18;
    jt[row * JT_STRIDE + col] = a.jt->pxy[row][col];
  }
  for (u32 i = tid; i < NB_JUMP * DW; i += WALK_BLOCK) jdist[i] = a.jt->dist[i / DW][i % DW];
  __syncthreads();

  const u64 T = (u64)blockIdx.x * WALK_BLOCK + tid;

  fe x[K], y[K];
  u32 delta[K][DW];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const u64 p = T + (u64)k * a.threads;  // always < cap: the host pads the planes
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u64 v = a.X[(u64)i * a.cap + p];
      w[2 * i] = (u32)v;
      w[2 * i + 1] = (u32)(v >> 32);
    }
    fe_from_words(x[k], w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u64 v = a.Y[(u64)i * a.cap + p];
      w[2 * i] = (u32)v;
      w[2 * i + 1] = (u32)(v >> 32);
    }
    fe_from_words(y[k], w);
#pragma unroll
    for (int i = 0; i < DW; ++i) delta[k][i] = 0;
  }

  // do-while (the host guarantees nb_run >= 1): a zero-trip bypass edge would make the allocator keep a
  // second copy of the whole loaded herd alive around the loop
  u32 run = 0;
#pragma unroll 1
  do {
    // Opaque copy of the thread index for everything address-like inside the loop: without it LLVM
    // hoists the (loop-invariant) DP-path and LDS addresses of all K kangaroos out of the loop and
    // keeps them in registers for the whole walk.
    u32 tidv = tid;
    asm volatile("" : "+v"(tidv));
    const u64 Tv = (u64)blockIdx.x * WALK_BLOCK + tidv;
    // ---- forward pass: prefix products of dx --------------------------------------------
    fe q;
    static_for<0, K>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      // keep the scheduler from interleaving different kangaroos: their bodies are independent, and
      // mixing them stretches live ranges until the state spills to scratch
      __builtin_amdgcn_sched_barrier(0);
      const u32 *row = jt + (x[k].l[0] & (NB_JUMP - 1)) * JT_STRIDE;
      fe jx, dx;
#pragma unroll
      for (int i = 0; i < 9; ++i) jx.l[i] = row[i];
      fe_sub_wide(dx, x[k], jx);
      if constexpr (k == 0) {
        q = dx;
        fe_weak(q);
      } else {
        fe_mul(q, dx, q);
      }
      if constexpr (k < K - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) pref[(k * 9 + i) * WALK_BLOCK + tidv] = q.l[i];
      }
    });
    // ---- one inversion per lane -----------------------------------------------------------
    fe u;
    fe_inv(u, q);
    // ---- backward pass: individual inverses, affine add, distance, DP ----------------------
    static_for_down<K - 1>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      __builtin_amdgcn_sched_barrier(0);
      const u32 j = x[k].l[0] & (NB_JUMP - 1);
      const u32 *row = jt + j * JT_STRIDE;
      fe jx, jy, dx, dxi;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        jx.l[i] = row[i];
        jy.l[i] = row[9 + i];
      }
      if constexpr (k > 0) {
        fe qp;
#pragma unroll
        for (int i = 0; i < 9; ++i) qp.l[i] = pref[((k - 1) * 9 + i) * WALK_BLOCK + tidv];
        fe_sub_wide(dx, x[k], jx);
        fe_mul(dxi, qp, u);   // (q_{k-1}) * (q_k)^-1 = dx_k^-1      (both weak)
        fe_mul(u, dx, u);     // (q_k)^-1 * dx_k = (q_{k-1})^-1
      } else {
        dxi = u;
      }
      fe dy, s, p2, rx, t, ry;
      fe_sub_wide(dy, y[k], jy);
      fe_mul(s, dy, dxi);
      fe_sqr(p2, s);
      fe_sub2_wide(rx, p2, jx, x[k]);
      fe_canon(rx);                 // x must be exact: it selects the next jump and is DP-tested
      fe_sub_wide(t, x[k], rx);
      fe_mul(ry, t, s);
      fe_sub_wide(ry, ry, y[k]);
      fe_weak(ry);
      x[k] = rx;
      y[k] = ry;
      delta_add<DW>(delta[k], jdist + j * DW);

      // distinguished point: (x.bits64[3] & dpMask) == 0   (math/gpu/compute.h:54)
      const u32 top_hi = (rx.l[8] << 8) | (rx.l[7] >> 21);
      const u32 top_lo = (rx.l[7] << 11) | (rx.l[6] >> 18);
      const u64 p = Tv + (u64)k * a.threads;
      const bool is_dp = ((top_hi & a.dp_mask_hi) | (top_lo & a.dp_mask_lo)) == 0 && p < a.n;
      const unsigned long long m = __ballot(is_dp);
      if (m != 0) {
        const u32 lane = __lane_id();
        const u32 leader = (u32)__ffsll(m) - 1;
        u32 base = 0;
        if (lane == leader) base = atomicAdd(a.out_count, (u32)__popcll(m));
        base = __shfl(base, (int)leader);
        if (is_dp) {
          const u32 pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < a.max_found) {
            DpItemDev *it = a.out + pos;
            u32 w[8];
            fe_to_words(w, rx);
#pragma unroll
            for (int i = 0; i < 8; ++i) it->x[i] = w[i];
            // distance = D[p] + delta  (plain 256-bit add, math/gpu/kernel.h:88-92)
            u64 c = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const u64 dv = a.D[(u64)i * a.cap + p];
              u64 lo = (u64)(u32)dv + (2 * i < DW ? delta[k][2 * i < DW ? 2 * i : 0] : 0u) + c;
              u64 hi = (dv >> 32) + (2 * i + 1 < DW ? delta[k][2 * i + 1 < DW ? 2 * i + 1 : 0] : 0u) + (lo >> 32);
              it->d[2 * i] = (u32)lo;
              it->d[2 * i + 1] = (u32)hi;
              c = hi >> 32;
            }
            it->kidx = p;
          }
        }
      }
    });
  } while (++run < a.nb_run);

  // ---- write the herd back ---------------------------------------------------------------------
  u32 tidw = tid;
  asm volatile("" : "+v"(tidw));  // keeps the store addresses from being computed (and spilled) before the loop
  const u64 Tw = (u64)blockIdx.x * WALK_BLOCK + tidw;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const u64 p = Tw + (u64)k * a.threads;
    u32 w[8];
    fe_to_words(w, x[k]);  // canonical already
#pragma unroll
    for (int i = 0; i < 4; ++i) a.X[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    fe yy = y[k];
    fe_canon(yy);
    fe_to_words(w, yy);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.Y[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u64 dv = a.D[(u64)i * a.cap + p];
      u64 lo = (u64)(u32)dv + (2 * i < DW ? delta[k][2 * i < DW ? 2 * i : 0] : 0u) + c;
      u64 hi = (dv >> 32) + (2 * i + 1 < DW ? delta[k][2 * i + 1 < DW ? 2 * i + 1 : 0] : 0u) + (lo >> 32);
      a.D[(u64)i * a.cap + p] = (u64)(u32)lo | (hi << 32);
      c = hi >> 32;
    }
  }
}

// LDS bytes for one workgroup of kg_walk_kernel<K, DW>
constexpr size_t walk_lds_bytes(int K, int DW) {
  return (size_t)(NB_JUMP * JT_STRIDE + NB_JUMP * DW + (K > 1 ? (K - 1) : 0) * 9 * WALK_BLOCK) * sizeof(u32);
}

}  // namespace kg
