// walk_kernel_rot.h — compact-code variant of the jump kernel: K kangaroos per lane stay in
// VGPR/AGPR registers, but only U of them ("the window", register slots 0..U-1) are touched by
// straight-line code; a rolled loop processes the window and then rotates the whole register-resident
// herd left by U slots.  Code size is U bodies (~10 KB each) instead of K, so K = 16 fits the
// instruction cache; the price is K*(18+DW)/U register moves per kangaroo and step.
//
// One fused pass per step (instead of a forward and a backward pass): while kangaroo id is being
// jumped with dx_id^-1 = u * P[id + dir], its NEXT dx' is multiplied into the running product of the
// next step and stored as P[id].  Consecutive passes therefore run in opposite directions over the
// kangaroo ids; the registers are reversed once per pass so that the window always sees them in order.
//
// See walk_common.h for the data layout.
#pragma once
#include "walk_common.h"

namespace kg {

template <int K, int N>
__device__ __forceinline__ void rot_left_fe(fe (&v)[K]) {
  fe t[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = v[i];
#pragma unroll
  for (int i = 0; i < K - N; ++i) v[i] = v[i + N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[K - N + i] = t[i];
}
template <int K>
__device__ __forceinline__ void reverse_fe(fe (&v)[K]) {
#pragma unroll
  for (int i = 0; i < K / 2; ++i) {
    fe t = v[i];
    v[i] = v[K - 1 - i];
    v[K - 1 - i] = t;
  }
}

template <int DW>
struct dlt {
  u32 w[DW];
};

template <int DW>
__device__ __forceinline__ void add_dist_256(u32 (&out)[8], const u64 *D, u64 cap, u64 p, const dlt<DW> &dl) {
  // distance = D[p] + delta  (plain 256-bit add, math/gpu/kernel.h:88-92)
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u64 dv = D[(u64)i * cap + p];
    u64 lo = (u64)(u32)dv + (2 * i < DW ? dl.w[2 * i < DW ? 2 * i : 0] : 0u) + c;
    u64 hi = (dv >> 32) + (2 * i + 1 < DW ? dl.w[2 * i + 1 < DW ? 2 * i + 1 : 0] : 0u) + (lo >> 32);
    out[2 * i] = (u32)lo;
    out[2 * i + 1] = (u32)hi;
    c = hi >> 32;
  }
}

template <int K, int U, int DW, int W>
__global__ void __launch_bounds__(WALK_BLOCK, W) kg_walk_kernel_rot(WalkArgs a) {
  static_assert(K % U == 0, "window must divide the herd per lane");
  extern __shared__ u32 lds[];
  u32 *jt = lds;                           // [64][JT_STRIDE]
  u32 *jdist = lds + NB_JUMP * JT_STRIDE;  // [64][DW]
  u32 *pref = jdist + NB_JUMP * DW;        // [K][9][WALK_BLOCK]

  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < NB_JUMP * 18; i += WALK_BLOCK) {
    u32 row = i / 18, col = i % 18;
    jt[row * JT_STRIDE + col] = a.jt->pxy[row][col];
  }
  for (u32 i = tid; i < NB_JUMP * DW; i += WALK_BLOCK) jdist[i] = a.jt->dist[i / DW][i % DW];
  __syncthreads();

  const u64 T = (u64)blockIdx.x * WALK_BLOCK + tid;

  fe x[K], y[K];
  dlt<DW> delta[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const u64 p = T + (u64)k * a.threads;
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
    for (int i = 0; i < DW; ++i) delta[k].w[i] = 0;
  }

  // ---- prologue: prefix products of the first step's dx, ascending ids ------------------------------
  fe acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);
#pragma unroll 1
  for (int blk = 0; blk < K / U; ++blk) {
    u32 tidv = tid;
    asm volatile("" : "+v"(tidv));
    static_for<0, U>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      __builtin_amdgcn_sched_barrier(0);
      const int id = blk * U + j;
      const u32 *row = jt + (x[j].l[0] & (NB_JUMP - 1)) * JT_STRIDE;
      fe jx, dx;
#pragma unroll
      for (int i = 0; i < 9; ++i) jx.l[i] = row[i];
      fe_sub_wide(dx, x[j], jx);
      fe_mul(acc, dx, acc);
#pragma unroll
      for (int i = 0; i < 9; ++i) pref[(id * 9 + i) * WALK_BLOCK + tidv] = acc.l[i];
    });
    rot_left_fe<K, U>(x);
  }
  // ids were visited ascending; the first step walks them descending
  reverse_fe<K>(x);
  reverse_fe<K>(y);

#ifdef KG_PROFILE
  u64 prof_inv = 0;
  const u64 prof_start = __builtin_amdgcn_s_memtime();
#endif
  u32 run = 0;
#pragma unroll 1
  do {
    // registers hold the ids in processing order: id(t) = asc ? t : K-1-t, dir = asc ? +1 : -1
    const bool asc = (run & 1) != 0;
    fe u;
#ifdef KG_PROFILE
    const u64 tp0 = __builtin_amdgcn_s_memtime();
#endif
    fe_inv(u, acc);
#ifdef KG_PROFILE
    const u64 tp1 = __builtin_amdgcn_s_memtime();
    prof_inv += tp1 - tp0;
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);

    // Software prefetch: the jump-table row and the stored partial product of kangaroo t+1 are read
    // from LDS while kangaroo t is being computed (one wave per SIMD has nothing else to hide the
    // LDS latency behind).  cur = operands of the kangaroo about to be processed.
    struct Pre {
      fe jx, jy, qp;
      u32 jmp;
    };
    auto prefetch = [&](Pre &pr, const fe &xs, int t, u32 tv) {
      pr.jmp = xs.l[0] & (NB_JUMP - 1);
      const u32 *row = jt + pr.jmp * JT_STRIDE;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        pr.jx.l[i] = row[i];
        pr.jy.l[i] = row[9 + i];
      }
      const int idn = asc ? t : K - 1 - t;
      int nb = asc ? idn + 1 : idn - 1;  // the id processed after this one; clamped for the last (unused there)
      nb = nb < 0 ? 0 : (nb > K - 1 ? K - 1 : nb);
#pragma unroll
      for (int i = 0; i < 9; ++i) pr.qp.l[i] = pref[(nb * 9 + i) * WALK_BLOCK + tv];
    };
    Pre cur;
    {
      u32 tv0 = tid;
      asm volatile("" : "+v"(tv0));
      prefetch(cur, x[0], 0, tv0);
    }

#pragma unroll 1
    for (int blk = 0; blk < K / U; ++blk) {
      u32 tidv = tid;
      asm volatile("" : "+v"(tidv));
      const u64 Tv = (u64)blockIdx.x * WALK_BLOCK + tidv;
      static_for<0, U>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        __builtin_amdgcn_sched_barrier(0);
        const int t = blk * U + j;
        const int id = asc ? t : K - 1 - t;
        // issue the loads for the next kangaroo first (slot j+1; slot U is the first of the next block)
        Pre nxt;
        prefetch(nxt, x[(j + 1) % K], t + 1 < K ? t + 1 : K - 1, tidv);
        __builtin_amdgcn_sched_barrier(0);
        const u32 jmp = cur.jmp;
        const fe jx = cur.jx, jy = cur.jy;
        fe dx, dxi;
        // dx_id^-1 = u * (product of the ids still to come in this pass)
        if (t < K - 1) {
          fe_sub_wide(dx, x[j], jx);
          fe_mul(dxi, cur.qp, u);
          fe_mul(u, dx, u);
        } else {
          dxi = u;
        }
        fe dy, s, p2, rx, tt, ry;
        fe_sub_wide(dy, y[j], jy);
        fe_mul(s, dy, dxi);
        fe_sqr(p2, s);
        fe_sub2_wide(rx, p2, jx, x[j]);
        fe_canon(rx);
        fe_sub_wide(tt, x[j], rx);
        fe_mul(ry, tt, s);
        fe_sub_wide(ry, ry, y[j]);
        fe_weak(ry);
        x[j] = rx;
        y[j] = ry;
        delta_add<DW>(delta[j].w, jdist + jmp * DW);

        // next step's dx and running product
        {
          const u32 *row2 = jt + (rx.l[0] & (NB_JUMP - 1)) * JT_STRIDE;
          fe jx2, dx2;
#pragma unroll
          for (int i = 0; i < 9; ++i) jx2.l[i] = row2[i];
          fe_sub_wide(dx2, rx, jx2);
          fe_mul(acc, dx2, acc);
#pragma unroll
          for (int i = 0; i < 9; ++i) pref[(id * 9 + i) * WALK_BLOCK + tidv] = acc.l[i];
        }

        // distinguished point: (x.bits64[3] & dpMask) == 0   (math/gpu/compute.h:54)
        const u32 top_hi = (rx.l[8] << 8) | (rx.l[7] >> 21);
        const u32 top_lo = (rx.l[7] << 11) | (rx.l[6] >> 18);
        const u64 p = Tv + (u64)id * a.threads;
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
              add_dist_256<DW>(w, a.D, a.cap, p, delta[j]);
#pragma unroll
              for (int i = 0; i < 8; ++i) it->d[i] = w[i];
              it->kidx = a.p_base + p;
            }
          }
        }
        cur = nxt;
      });
      rot_left_fe<K, U>(x);
      rot_left_fe<K, U>(y);
      {
        dlt<DW> tmp[U];
#pragma unroll
        for (int i = 0; i < U; ++i) tmp[i] = delta[i];
#pragma unroll
        for (int i = 0; i < K - U; ++i) delta[i] = delta[i + U];
#pragma unroll
        for (int i = 0; i < U; ++i) delta[K - U + i] = tmp[i];
      }
    }
    // next pass runs the other way
    reverse_fe<K>(x);
    reverse_fe<K>(y);
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {
      dlt<DW> tmp = delta[i];
      delta[i] = delta[K - 1 - i];
      delta[K - 1 - i] = tmp;
    }
  } while (++run < a.nb_run);

#ifdef KG_PROFILE
  if (blockIdx.x == 0 && (tid & 63) == 0)
    printf("[kg profile] wave %u: loop %llu cycles, inversion %llu cycles (%u steps)\n", tid >> 6,
           (unsigned long long)(__builtin_amdgcn_s_memtime() - prof_start), (unsigned long long)prof_inv, a.nb_run);
#endif
  // ---- write the herd back: slot s holds id s if the last pass was descending (it ended with a reversal) ----
  const bool natural = ((a.nb_run - 1) & 1) == 0;
  u32 tidw = tid;
  asm volatile("" : "+v"(tidw));
  const u64 Tw = (u64)blockIdx.x * WALK_BLOCK + tidw;
#pragma unroll
  for (int sIdx = 0; sIdx < K; ++sIdx) {
    const int id = natural ? sIdx : K - 1 - sIdx;
    const u64 p = Tw + (u64)id * a.threads;
    u32 w[8];
    fe_to_words(w, x[sIdx]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.X[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    fe yy = y[sIdx];
    fe_canon(yy);
    fe_to_words(w, yy);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.Y[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    add_dist_256<DW>(w, a.D, a.cap, p, delta[sIdx]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.D[(u64)i * a.cap + p] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
  }
}

constexpr size_t walk_rot_lds_bytes(int K, int DW) {
  return (size_t)(NB_JUMP * JT_STRIDE + NB_JUMP * DW + K * 9 * WALK_BLOCK) * sizeof(u32);
}

}  // namespace kg
