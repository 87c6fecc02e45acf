// herd_kernel.h — kangaroo start points on the GPU: P_i = d_i * G (+ Q for wild kangaroos).
//
// Replaces the CPU loop of Kangaroo::CreateHerd (core/kangaroo.cpp:521-573: Secp256K1::ComputePublicKeys +
// AddDirect, about 72 k points/s per host core) for herds of millions of kangaroos.
//
// Fixed-base comb, 32 windows of 8 bits.  The host builds tab[w][j] = (j * 2^(8w) + c_w) * G and the start
// point S = -(sum c_w) * G (host_secp.cpp: kgh_comb_table), so every window is one affine addition
// P += tab[w][digit_w(d)] with no special cases; wild kangaroos start from S + Q instead of S.  The 32
// additions of a point are 32 "steps" with a per-step table of 256 entries (18 KB, staged in LDS); the
// inverse of each addition is shared by the K points of a lane (Montgomery trick, two passes per window),
// the points live in a workspace in HBM: x, y as 9 limbs each [slot][18][lane] and the prefix products
// [slot][9][lane].  Field arithmetic: fe29.h / modinv30.h, as in the jump kernels.
#pragma once
#include "walk_common.h"

namespace kg {

constexpr int HERD_WINDOWS = 32;
constexpr int HERD_TAB_STRIDE = 19;  // LDS row: 18 limbs + 1 pad

struct HerdArgs {
  const u32 *tab;    // [32][256][18]
  const u64 *k;      // n scalars, 4 limbs each (host order)
  u64 n;             // points
  u32 lanes;         // total threads; lane T owns points T + slot * lanes
  u32 K;             // slots per lane
  u32 *ws;           // [K][27][lanes]
  u64 *X, *Y;        // output planes, limb i of point p at [i * cap + p]
  u64 cap;
  u32 start[2][18];  // start point for even / odd p (tame: S, wild: S + Q), 9+9 limbs
};

__global__ void __launch_bounds__(WALK_BLOCK, 2) kg_herd_kernel(HerdArgs a) {
  __shared__ u32 tab[256 * HERD_TAB_STRIDE];
  const u32 tid = threadIdx.x;
  const u32 T = blockIdx.x * WALK_BLOCK + tid;
  const u32 lanes = a.lanes, K = a.K;
  u32 *wsx = a.ws + T;  // row r of slot s: wsx[(s * 27 + r) * lanes]

  // start points
  for (u32 s = 0; s < K; ++s) {
    const u64 p = (u64)T + (u64)s * lanes;
    const u32 par = (u32)(p & 1);
#pragma unroll
    for (int i = 0; i < 18; ++i) wsx[(u64)(s * 27 + i) * lanes] = a.start[par][i];
  }

#pragma unroll 1
  for (int w = 0; w < HERD_WINDOWS; ++w) {
    __syncthreads();
    for (u32 i = tid; i < 256 * 18; i += WALK_BLOCK) tab[(i / 18) * HERD_TAB_STRIDE + i % 18] = a.tab[(u64)w * 256 * 18 + i];
    __syncthreads();

    auto digit_of = [&](u32 s) {
      u64 p = (u64)T + (u64)s * lanes;
      p = p < a.n ? p : a.n - 1;  // padding lanes replay the last point, their result is not stored
      return (u32)(a.k[4 * p + (w >> 3)] >> (8 * (w & 7))) & 255u;
    };

    // forward: prefix products of dx = tx - x
    fe acc;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc.l[i] = (i == 0);
#pragma unroll 1
    for (u32 s = 0; s < K; ++s) {
      const u32 *row = tab + digit_of(s) * HERD_TAB_STRIDE;
      fe x, tx, dx;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        x.l[i] = wsx[(u64)(s * 27 + i) * lanes];
        tx.l[i] = row[i];
      }
      fe_sub_wide(dx, tx, x);
      fe_mul(acc, dx, acc);
#pragma unroll
      for (int i = 0; i < 9; ++i) wsx[(u64)(s * 27 + 18 + i) * lanes] = acc.l[i];
    }
    fe u;
    fe_inv(u, acc);
    // backward: dx_s^-1 = u * prefix[s-1]; affine addition
#pragma unroll 1
    for (u32 sr = 0; sr < K; ++sr) {
      const u32 s = K - 1 - sr;
      const u32 *row = tab + digit_of(s) * HERD_TAB_STRIDE;
      fe x, y, tx, ty, dx, dxi;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        x.l[i] = wsx[(u64)(s * 27 + i) * lanes];
        y.l[i] = wsx[(u64)(s * 27 + 9 + i) * lanes];
        tx.l[i] = row[i];
        ty.l[i] = row[9 + i];
      }
      fe_sub_wide(dx, tx, x);
      if (s > 0) {
        fe q;
#pragma unroll
        for (int i = 0; i < 9; ++i) q.l[i] = wsx[(u64)((s - 1) * 27 + 18 + i) * lanes];
        fe_mul(dxi, q, u);
        fe_mul(u, dx, u);
      } else {
        dxi = u;
      }
      fe dy, sl, p2, rx, tt, ry;
      fe_sub_wide(dy, ty, y);
      fe_mul(sl, dy, dxi);
      fe_sqr(p2, sl);
      fe_sub2_wide(rx, p2, tx, x);
      fe_weak(rx);
      fe_sub_wide(tt, x, rx);
      fe_mul(ry, tt, sl);
      fe_sub_wide(ry, ry, y);
      fe_weak(ry);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        wsx[(u64)(s * 27 + i) * lanes] = rx.l[i];
        wsx[(u64)(s * 27 + 9 + i) * lanes] = ry.l[i];
      }
    }
  }

  // canonical coordinates into the output planes
#pragma unroll 1
  for (u32 s = 0; s < K; ++s) {
    const u64 p = (u64)T + (u64)s * lanes;
    if (p >= a.n) continue;
    fe x, y;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      x.l[i] = wsx[(u64)(s * 27 + i) * lanes];
      y.l[i] = wsx[(u64)(s * 27 + 9 + i) * lanes];
    }
    fe_canon(x);
    fe_canon(y);
    u32 w8[8];
    fe_to_words(w8, x);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.X[(u64)i * a.cap + p] = (u64)w8[2 * i] | ((u64)w8[2 * i + 1] << 32);
    fe_to_words(w8, y);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.Y[(u64)i * a.cap + p] = (u64)w8[2 * i] | ((u64)w8[2 * i + 1] << 32);
  }
}

// padding entries [n, cap) of the engine planes replay kangaroo 0 (they are walked but never reported)
__global__ void kg_pad_kernel(u64 *X, u64 *Y, u64 *D, u64 cap, u64 n) {
  const u64 p = n + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cap) return;
  for (int i = 0; i < 4; ++i) {
    X[(u64)i * cap + p] = X[(u64)i * cap];
    Y[(u64)i * cap + p] = Y[(u64)i * cap];
    D[(u64)i * cap + p] = D[(u64)i * cap];
  }
}

}  // namespace kg
