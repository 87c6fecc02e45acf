// modinv30.h — modular inverse mod p for gfx950 lanes, without divergence inside a wavefront.
//
// Replaces the reference's data-dependent 62-bit divstep loop (math/gpu/kernel.h:565-655,
// math/cpu/intmod.cpp:153-292), which diverges across a wavefront, with the Bernstein-Yang "safegcd"
// iteration on signed 30-bit limbs.  Three forms (fe_inv_t<MODE>):
//   MODE 2 (the default, what the kernels run): batches of up to 30 divsteps, each batch KG_INV_NITER
//     branch-free iterations (divsteps_30_fix); the wavefront loops until its slowest lane is done
//     (about 21 batches), with a hard cap of KG_INV_MAX_BATCHES so that corrupted state can only give a
//     wrong inverse (caught by the parity and invariant checks), never an endless loop;
//   MODE 1: variable-time batches with a per-lane loop (divsteps_30_var);
//   MODE 0: the fixed-length constant-time form, 20 batches of 30 divsteps (600 >= the 590 divsteps proven
//     sufficient for 256-bit inputs with the delta = 1/2 start).
// Every lane executes the same instruction stream, so 64 different inversions cost one wave-pass.
// Output: canonical a^-1 in [0,p); 0 for a = 0 (mod p) — the same contract as the reference's
// _ModInv / Int::ModInv.
//
// Credit: the limb layout, divsteps_30 / divsteps_30_var, the trans2x2 matrix, update_de_30 / update_fg_30
// and normalize_30 follow the modinv32 implementation of bitcoin-core's libsecp256k1 (src/modinv32_impl.h,
// MIT licence, Pieter Wuille et al.), itself an implementation of Bernstein & Yang, "Fast constant-time gcd
// computation and modular inversion" (2019).  divsteps_30_fix, update_defg_30 and the SIMT loop are this
// repository's own.
#pragma once
#include "fe29.h"

namespace kg {

constexpr i32 M30 = 0x3FFFFFFF;
constexpr u32 P_INV30 = 0x2DDACACFu;  // p^-1 mod 2^30

struct s30 {
  i32 v[9];
};
struct trans2x2 {
  i32 u, v, q, r;
};

// p = 2^256 - 2^32 - 977 in signed 30-bit limbs: {-977, -4, 0, 0, 0, 0, 0, 0, 65536}
KG_HD i32 p30(int i) { return i == 0 ? -0x3D1 : i == 1 ? -4 : i == 8 ? 65536 : 0; }

// c + a * b on signed 32-bit operands.  hipcc lowers the C expression to v_mad_u64_u32 plus v_mul_lo_u32
// sign corrections (3 instructions per product); the hardware has the signed form.
KG_HD i64 mad_i64(i32 a, i32 b, i64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  // accumulate in place; the carry-out goes to a scratch SGPR pair so that neighbouring products do
  // not serialise on VCC
  u64 carry;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(carry) : "v"(a), "v"(b));
  return c;
#else
  return c + (i64)a * b;
#endif
}

// true if the predicate holds in any lane of the wavefront (the host build has one "lane")
KG_HD bool kg_any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(p) != 0;
#else
  return p;
#endif
}

// 30 divsteps on the low words; returns the new zeta and the transition matrix (scaled 2^30)
KG_HD i32 divsteps_30(i32 zeta, u32 f0, u32 g0, trans2x2 &t) {
  u32 u = 1, v = 0, q = 0, r = 1;
  u32 f = f0, g = g0;
#pragma unroll 5
  for (int i = 0; i < 30; ++i) {
    u32 c1 = (u32)(zeta >> 31);     // zeta < 0
    u32 c2 = (u32)0 - (g & 1u);     // g odd
    u32 x = (f ^ c1) - c1;          // conditionally negated f, u, v
    u32 y = (u ^ c1) - c1;
    u32 z = (v ^ c1) - c1;
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;                       // zeta < 0 and g odd: swap roles
    zeta = (i32)((u32)zeta ^ c1) - 1;
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (i32)u;
  t.v = (i32)v;
  t.q = (i32)q;
  t.r = (i32)r;
  return zeta;
}

// Variable-time batch of 30 divsteps (the delta = 1 form, eta = -delta): every iteration strips the
// trailing zeros of g at once and then cancels up to 8 more low bits with one multiple of f.  Lanes
// take different iteration counts; the loop runs until the slowest lane of the wavefront is done
// (about 11.5 iterations against 30 fixed steps, measured), finished lanes are masked off.
KG_HD i32 divsteps_30_var(i32 eta, u32 f0, u32 g0, trans2x2 &t) {
  u32 u = 1, v = 0, q = 0, r = 1;
  u32 f = f0, g = g0;
  i32 i = 30;
  for (;;) {
    const u32 zeros = (u32)__builtin_ctz(g | (0xFFFFFFFFu << i));  // sentinel: never more than i
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= (i32)zeros;
    i -= (i32)zeros;
    if (i == 0) break;
    if (eta < 0) {  // swap roles: (f, g) <- (g, -f)
      u32 tmp;
      eta = -eta;
      tmp = f; f = g; g = 0u - tmp;
      tmp = u; u = q; q = 0u - tmp;
      tmp = v; v = r; r = 0u - tmp;
    }
    const i32 limit = (eta + 1) > i ? i : (eta + 1);
    const u32 m = (0xFFFFFFFFu >> (32 - limit)) & 255u;
    // f^-1 mod 2^8 by Newton: (3f)^2 is correct to 5 bits, one step gives 10
    u32 fi = (f * 3u) ^ 2u;
    fi *= 2u - f * fi;
    const u32 w = (g * (0u - fi)) & m;  // multiple of f that clears the low min(limit, 8) bits of g
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t.u = (i32)u;
  t.v = (i32)v;
  t.q = (i32)q;
  t.r = (i32)r;
  return eta;
}

// Branch-free form of the variable-time batch for SIMT: a FIXED number of iterations, each consuming
// however many of the batch's 30 bits it can (trailing zeros of g, then up to 5 more cleared with one
// multiple of f), no data-dependent control flow and no v_cndmask (slow on gfx950, measured 22 cycles).
// Bits a lane has not consumed after NITER iterations are handed to the next batch: scaling all four
// matrix entries by 2^rem makes the division by 2^30 in update_fg/update_de a division by 2^consumed.
// Same divstep sequence as divsteps_30_var, only the batch boundaries differ.
template <int NITER>
KG_HD i32 divsteps_30_fix(i32 eta, u32 f0, u32 g0, trans2x2 &t) {
  u32 u = 1, v = 0, q = 0, r = 1;
  u32 f = f0, g = g0;
  i32 rem = 30;  // bits left in this batch (0: batch full, the remaining iterations are no-ops)
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const u32 zeros = (u32)__builtin_ctz(g | (0xFFFFFFFFu << rem));  // sentinel: never more than rem
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= (i32)zeros;
    rem -= (i32)zeros;
    // swap roles when eta < 0 and the batch is not full: (f, g) <- (g, -f); as masks
    const u32 cm = (u32)(((i32)((u32)eta & ~(u32)(rem - 1))) >> 31);
    const u32 nf = (g & cm) | (f & ~cm);
    const u32 ng = (((f & cm) | (g & ~cm)) ^ cm) - cm;
    const u32 nu = (q & cm) | (u & ~cm);
    const u32 nq = (((u & cm) | (q & ~cm)) ^ cm) - cm;
    const u32 nv = (r & cm) | (v & ~cm);
    const u32 nr = (((v & cm) | (r & ~cm)) ^ cm) - cm;
    f = nf; g = ng; u = nu; q = nq; v = nv; r = nr;
    eta = (i32)(((u32)eta ^ cm) - cm);
    // clear the low min(eta + 1, bits left, 5) bits of g with a multiple of f
    i32 lim = rem < 5 ? rem : 5;
    lim = eta + 1 < lim ? eta + 1 : lim;
    lim = lim < 0 ? 0 : lim;  // eta can only be negative here when the batch is full (rem = 0)
    const u32 fi = (f * 3u) ^ 2u;  // f^-1 mod 32
    const u32 w = (0u - g * fi) & ((1u << lim) - 1u);
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t.u = (i32)(u << rem);
  t.v = (i32)(v << rem);
  t.q = (i32)(q << rem);
  t.r = (i32)(r << rem);
  return eta;
}

// (f,g) <- t * (f,g) / 2^30
KG_HD void update_fg_30(s30 &f, s30 &g, const trans2x2 &t) {
  const i32 u = t.u, v = t.v, q = t.q, r = t.r;  // 32-bit x 32-bit -> v_mad_i64_i32
  i64 cf = mad_i64(u, f.v[0], mad_i64(v, g.v[0], 0));
  i64 cg = mad_i64(q, f.v[0], mad_i64(r, g.v[0], 0));
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cf = mad_i64(u, f.v[i], mad_i64(v, g.v[i], cf));
    cg = mad_i64(q, f.v[i], mad_i64(r, g.v[i], cg));
    f.v[i - 1] = (i32)cf & M30;
    cf >>= 30;
    g.v[i - 1] = (i32)cg & M30;
    cg >>= 30;
  }
  f.v[8] = (i32)cf;
  g.v[8] = (i32)cg;
}

// (d,e) <- t * (d,e) / 2^30 (mod p), keeping both in (-2p, p)
KG_HD void update_de_30(s30 &d, s30 &e, const trans2x2 &t) {
  const i32 u = t.u, v = t.v, q = t.q, r = t.r;
  i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;
  i32 md = (u & sd) + (v & se);
  i32 me = (q & sd) + (r & se);
  i64 cd = mad_i64(u, d.v[0], mad_i64(v, e.v[0], 0));
  i64 ce = mad_i64(q, d.v[0], mad_i64(r, e.v[0], 0));
  md -= (i32)((P_INV30 * (u32)cd + (u32)md) & (u32)M30);
  me -= (i32)((P_INV30 * (u32)ce + (u32)me) & (u32)M30);
  cd = mad_i64(p30(0), md, cd);
  ce = mad_i64(p30(0), me, ce);
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cd = mad_i64(u, d.v[i], mad_i64(v, e.v[i], cd));
    ce = mad_i64(q, d.v[i], mad_i64(r, e.v[i], ce));
    if (i == 1) {  // p30(1) = -4
      cd -= (i64)md << 2;
      ce -= (i64)me << 2;
    }
    if (i == 8) {  // p30(8) = 2^16
      cd += (i64)md << 16;
      ce += (i64)me << 16;
    }
    d.v[i - 1] = (i32)cd & M30;
    cd >>= 30;
    e.v[i - 1] = (i32)ce & M30;
    ce >>= 30;
  }
  d.v[8] = (i32)cd;
  e.v[8] = (i32)ce;
}

// Same product, but the instruction keeps its place in the source order (volatile): used by the fused
// update below to interleave four independent carry chains by hand — left to itself the scheduler
// emits each chain back to back and a lone wave per SIMD then waits out every result latency.
KG_HD i64 mad_i64_ord(i32 a, i32 b, i64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 carry;
  asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(carry) : "v"(a), "v"(b));
  return c;
#else
  return c + (i64)a * b;
#endif
}

// update_de_30 and update_fg_30 in one pass over the limbs, the four chains (d, e, f, g) round-robin
KG_HD void update_defg_30(s30 &d, s30 &e, s30 &f, s30 &g, const trans2x2 &t) {
  const i32 u = t.u, v = t.v, q = t.q, r = t.r;
  const i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;
  i32 md = (u & sd) + (v & se);
  i32 me = (q & sd) + (r & se);
  i64 cd = mad_i64_ord(u, d.v[0], 0);
  i64 ce = mad_i64_ord(q, d.v[0], 0);
  i64 cf = mad_i64_ord(u, f.v[0], 0);
  i64 cg = mad_i64_ord(q, f.v[0], 0);
  cd = mad_i64_ord(v, e.v[0], cd);
  ce = mad_i64_ord(r, e.v[0], ce);
  cf = mad_i64_ord(v, g.v[0], cf);
  cg = mad_i64_ord(r, g.v[0], cg);
  md -= (i32)((P_INV30 * (u32)cd + (u32)md) & (u32)M30);
  me -= (i32)((P_INV30 * (u32)ce + (u32)me) & (u32)M30);
  cf >>= 30;
  cg >>= 30;
  cd = mad_i64_ord(p30(0), md, cd);
  ce = mad_i64_ord(p30(0), me, ce);
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cd = mad_i64_ord(u, d.v[i], cd);
    ce = mad_i64_ord(q, d.v[i], ce);
    cf = mad_i64_ord(u, f.v[i], cf);
    cg = mad_i64_ord(q, f.v[i], cg);
    cd = mad_i64_ord(v, e.v[i], cd);
    ce = mad_i64_ord(r, e.v[i], ce);
    cf = mad_i64_ord(v, g.v[i], cf);
    cg = mad_i64_ord(r, g.v[i], cg);
    if (i == 1) {  // p30(1) = -4
      cd -= (i64)md << 2;
      ce -= (i64)me << 2;
    }
    if (i == 8) {  // p30(8) = 2^16
      cd += (i64)md << 16;
      ce += (i64)me << 16;
    }
    d.v[i - 1] = (i32)cd & M30;
    e.v[i - 1] = (i32)ce & M30;
    f.v[i - 1] = (i32)cf & M30;
    g.v[i - 1] = (i32)cg & M30;
    cd >>= 30;
    ce >>= 30;
    cf >>= 30;
    cg >>= 30;
  }
  d.v[8] = (i32)cd;
  e.v[8] = (i32)ce;
  f.v[8] = (i32)cf;
  g.v[8] = (i32)cg;
}

// bring d from (-2p, p) into [0, p), negating first when sign < 0
KG_HD void normalize_30(s30 &r, i32 sign) {
  i32 cond_add = r.v[8] >> 31;
  i32 cond_neg = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    i32 x = r.v[i] + (p30(i) & cond_add);
    r.v[i] = (x ^ cond_neg) - cond_neg;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
  cond_add = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] += p30(i) & cond_add;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
}

// a: any weak/wide field element.  r: canonical inverse (weak-compatible), 0 if a == 0 mod p.
// MODE 0: fixed 20 x 30 divsteps (constant time).  MODE 1: variable-time batches with a per-lane loop.
// MODE 2: variable-time batches of KG_INV_NITER branch-free iterations (the SIMT form; default).
#ifndef KG_INV_NITER
#define KG_INV_NITER 8
#endif
// Batches the MODE 2 loop may run: the delta = 1 iteration needs at most 724 divsteps for 256-bit inputs
// (25 full batches); every branch-free iteration consumes at least one bit, so a batch of 8 consumes at least 8
// and usually all 30: about 21 batches in practice (tools/ubench_inv.hip), never more than 724 / 8 = 91.
#ifndef KG_INV_MAX_BATCHES
#define KG_INV_MAX_BATCHES 96
#endif
template <int MODE, int NITER = KG_INV_NITER>
KG_HD void fe_inv_t(fe &r, const fe &a_in) {
  fe a = a_in;
  fe_canon(a);
  s30 d, e, f, g;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    d.v[i] = 0;
    e.v[i] = 0;
    f.v[i] = p30(i);
  }
  e.v[0] = 1;
  // 29-bit limbs -> 30-bit limbs: limb i starts at bit 30 i = 29 i + i
#pragma unroll
  for (int i = 0; i < 8; ++i) g.v[i] = (i32)(((a.l[i] >> i) | (a.l[i + 1] << (29 - i))) & (u32)M30);
  g.v[8] = (i32)(a.l[8] >> 8);
  i32 zeta = -1;
  if (MODE == 2) {
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nz |= (u32)g.v[i];
#pragma unroll 1
    for (int batch = 0; batch < KG_INV_MAX_BATCHES && kg_any_lane(nz != 0); ++batch) {
      // lanes that are done keep iterating harmlessly (g = 0 leaves d, f alone)
      trans2x2 t;
      zeta = divsteps_30_fix<NITER>(zeta, (u32)f.v[0], (u32)g.v[0], t);
      update_defg_30(d, e, f, g, t);
      nz = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) nz |= (u32)g.v[i];
    }
  } else if (MODE == 1) {
    // delta = 1 variant: at most 724 divsteps (25 batches) in theory, 17-19 batches in practice
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nz |= (u32)g.v[i];
#pragma unroll 1
    while (nz != 0) {
      trans2x2 t;
      zeta = divsteps_30_var(zeta, (u32)f.v[0], (u32)g.v[0], t);
      update_de_30(d, e, t);
      update_fg_30(f, g, t);
      nz = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) nz |= (u32)g.v[i];
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
      trans2x2 t;
      zeta = divsteps_30(zeta, (u32)f.v[0], (u32)g.v[0], t);
      update_de_30(d, e, t);
      update_fg_30(f, g, t);
    }
  }
  // g == 0 now, f == +-1 (or +-p when a == 0, in which case d == 0)
  normalize_30(d, f.v[8]);
  r.l[0] = (u32)d.v[0] & M29;
#pragma unroll
  for (int i = 1; i < 9; ++i)
    r.l[i] = (((u32)d.v[i - 1] >> (30 - i)) | ((u32)d.v[i] << i)) & M29;
}

#ifndef KG_INV_VARTIME
#define KG_INV_VARTIME 2
#endif
KG_HD void fe_inv(fe &r, const fe &a) { fe_inv_t<KG_INV_VARTIME>(r, a); }

}  // namespace kg
