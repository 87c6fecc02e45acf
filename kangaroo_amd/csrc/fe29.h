// fe29.h — secp256k1 field arithmetic for gfx950, 9 limbs of 29 bits in 32-bit VGPRs.
//
// Why not 4x64 / 8x32 saturated limbs: measured on MI355X (profiles/ubench_instr_table_r01.txt) a carry
// instruction (v_add_co/v_addc_co_u32) costs as much issue time as a v_mad_u64_u32 (4.3-5
// cycles per wave64 against 2.5 for a plain 32-bit op) and every VCC producer->consumer pair
// needs two wait states.  With 29-bit limbs the 81 partial products of a 256x256 multiply are
// accumulated by v_mad_u64_u32 straight into 64-bit column sums that cannot overflow
// (9 * 2^31 * 2^29 < 2^64), no carry flag is ever used, and subtraction is limb-wise.
//
// Values: v = sum l[i] * 2^(29 i).  Three states are used:
//   canonical : v in [0,p), l[0..7] < 2^29, l[8] < 2^24
//   weak      : l[0] < 2^29 + 2^18, l[1] < 2^29 + 2^11, l[2] < 2^29 + 2^16, l[3..7] < 2^29 (+1), l[8] <= 2^24;  v < 2^256 + 2^62
//   wide      : l[i] < 2^32 (a limb-wise a + c*p - b that has not been carried);   v < 2^259
// fe_mul takes (wide, weak) and returns weak; fe_sqr takes weak and returns weak.
//
// The reference computes the same residues with a lazy two-fold reduction
// (math/cpu/intmod.cpp:545-593, math/gpu/kernel.h:659-749) whose 256-bit output is canonical
// unless it lands in [p, 2^256) (density 2^-223); fe_canon() returns the canonical
// representative, so results are bit-identical to the reference wherever the reference's
// own value is canonical (see DESIGN.md "lazy reduction").
//
// The file is plain C++ so that tests can also compile it for the host (tests/fe_host.cpp);
// the product only ever runs it inside the HIP kernels.
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__) || defined(__HIP__)
#define KG_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define KG_HD static inline
#endif

namespace kg {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr u32 M29 = 0x1FFFFFFFu;
constexpr u32 FE_R0 = 0x7A20u;  // 2^261 = 2^37 + 0x7A20 (mod p): low part
constexpr u32 FE_R1 = 256u;     //   2^37 = 256 * 2^29

struct fe {
  u32 l[9];
};

// 2p and 4p with every limb lifted so that limb-wise (a + B - b) cannot go negative
#define KG_B2_INIT {0x5ffff85eu, 0x5fffffedu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x1fffffdu}
#define KG_B4_INIT {0x9ffff0bcu, 0x9fffffdbu, 0x9ffffffbu, 0x9ffffffbu, 0x9ffffffbu, 0x9ffffffbu, 0x9ffffffbu, 0x9ffffffbu, 0x3fffffbu}

KG_HD u32 fe_b2(int i) {
  return i == 0 ? 0x5ffff85eu : i == 1 ? 0x5fffffedu : i == 8 ? 0x1fffffdu : 0x5ffffffdu;
}
KG_HD u32 fe_b4(int i) {
  return i == 0 ? 0x9ffff0bcu : i == 1 ? 0x9fffffdbu : i == 8 ? 0x3fffffbu : 0x9ffffffbu;
}

// ---- conversion: eight 32-bit words (little endian) <-> nine 29-bit limbs ------------
KG_HD void fe_from_words(fe &r, const u32 w[8]) {
  r.l[0] = w[0] & M29;
  r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
  r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
  r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
  r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
  r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
  r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
  r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
  r.l[8] = w[7] >> 8;
}
// input must be canonical (or at least fully carried with l[8] < 2^24)
KG_HD void fe_to_words(u32 w[8], const fe &a) {
  w[0] = a.l[0] | (a.l[1] << 29);
  w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
  w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
  w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
  w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
  w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
  w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
  w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
}

// ---- limb-wise subtraction, no carries: r = a - b (+2p), result is WIDE ----------------
// a: limbs < 2^29 + 2^18 (weak);  b: limbs <= 2^30 - 3 and b.l[8] <= 2^25 - 3
KG_HD void fe_sub_wide(fe &r, const fe &a, const fe &b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + fe_b2(i) - b.l[i];
}
// r = a - b - c (+4p), WIDE.  a weak; b, c weak
KG_HD void fe_sub2_wide(fe &r, const fe &a, const fe &b, const fe &c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + fe_b4(i) - b.l[i] - c.l[i];
}

// The same subtractions with the subtrahend already lifted and negated: c = 2p' - b (resp. 4p' - b) limb by limb,
// prepared once for constants that are subtracted over and over (the jump table).  Results are WIDE, limb for limb
// the values fe_sub_wide(r, a, b) and fe_sub2_wide(r, a, b, x) give.
KG_HD void fe_add_lifted(fe &r, const fe &a, const fe &c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + c.l[i];
}
KG_HD void fe_add_lifted_sub(fe &r, const fe &a, const fe &c4, const fe &x) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + c4.l[i] - x.l[i];
}

// ---- carry + fold bit 256 and above: wide -> weak ------------------------------------------
KG_HD void fe_weak(fe &r) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 t = r.l[i] + c;  // l[i] < 2^32 - 8, c < 8
    r.l[i] = t & M29;
    c = t >> 29;
  }
  u32 t8 = r.l[8] + c;
  u32 v = t8 >> 24;  // multiples of 2^256 -> * (2^32 + 977)
  r.l[8] = t8 & 0xFFFFFFu;
  r.l[0] += v * 977u;  // v < 2^8 : stays below 2^29 + 2^18
  r.l[1] += v << 3;    // 2^32 = 8 * 2^29
}

// ---- full reduction to the canonical representative: any wide/weak -> canonical ----------
// the rare part: r has been through fe_weak; carries everything once more and handles bit 256 / values in [p, 2^256)
KG_HD void fe_canon_rare(fe &r) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 t = r.l[i] + c;
    r.l[i] = t & M29;
    c = t >> 29;
  }
  r.l[8] += c;
  // v in [0, 2^256 + eps).  Two cases remain: bit 256 set, or v in [p, 2^256).
  u32 all = r.l[2] & r.l[3] & r.l[4] & r.l[5] & r.l[6] & r.l[7];
  bool top = (r.l[8] >> 24) != 0;
  bool maybe_ge_p = (all == M29) && (r.l[8] == 0xFFFFFFu) && (r.l[1] >= 0x1ffffff7u);
  if (top || maybe_ge_p) {
    // add K = 2^32 + 977; if that carries out of bit 256 then v >= p and v - p = v + K - 2^256
    fe t = r;
    t.l[0] += 977u;
    t.l[1] += 8u;
    u32 cc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32 s = t.l[i] + cc;
      t.l[i] = s & M29;
      cc = s >> 29;
    }
    t.l[8] += cc;
    if (t.l[8] >> 24) {  // v + K >= 2^256  <=>  v >= p
      t.l[8] &= 0xFFFFFFu;
      r = t;  // v - p (v < 2p always holds here)
    }
  }
}
KG_HD void fe_canon(fe &r) {
  fe_weak(r);
  // now limbs 2..7 < 2^29 and limb 8 < 2^24; only limbs 0 and 1 can exceed 29 bits (by the fold of bit 256 and up).
  // Carry those two; anything further (a carry running on into limb 3, or a value in [p, 2^256)) is the rare part.
  r.l[1] += r.l[0] >> 29;
  r.l[0] &= M29;
  r.l[2] += r.l[1] >> 29;
  r.l[1] &= M29;
  const u32 all = r.l[2] & r.l[3] & r.l[4] & r.l[5] & r.l[6] & r.l[7];
  const bool rare = (r.l[2] >> 29) != 0 || ((all == M29) && (r.l[8] == 0xFFFFFFu) && (r.l[1] >= 0x1ffffff7u));
  if (__builtin_expect(rare, 0)) fe_canon_rare(r);
}

// ---- settle: weak -> fully carried 29-bit limbs with value < 2^256 (not necessarily < p), same residue --------------
// For values whose excess over 29 bits sits in limbs FIRST..LAST only (fe_weak output: limbs 0 and 1; fe_mul / fe_sqr
// output: limb 2).  Those limbs are carried; if the carry would run on past limb LAST + 1 (about once in 2^22 for a
// product, far rarer for fe_weak output) the value is canonicalised instead.  The result can be packed into eight
// 32-bit words with fe_to_words and comes back as a valid weak value with fe_from_words.
template <int FIRST, int LAST>
KG_HD void fe_settle(fe &r) {
  static_assert(FIRST <= LAST && LAST < 7, "carry must end below limb 8");
#pragma unroll
  for (int i = FIRST; i <= LAST; ++i) {
    r.l[i + 1] += r.l[i] >> 29;
    r.l[i] &= M29;
  }
  if (__builtin_expect((r.l[LAST + 1] >> 29) != 0, 0)) fe_canon(r);
}

// ---- multiplication: a wide (limbs < 2^31.1, value < 2^259), b weak -> weak ----------------
// acc + a * b, one v_mad_u64_u32.  Written as an asm statement on the device so that the column sums below
// stay single accumulate chains that start from the carry of the previous column: left as C, the
// compiler reassociates the 64-bit sums (chains started from zero, carries and the 2^37 term added with
// separate 64-bit adds and shifts), about 30 % more instructions per multiplication.
KG_HD u64 mad_u64(u32 a, u32 b, u64 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b));
  return acc;
#else
  return acc + (u64)a * b;
#endif
}
// the same with a small constant (kept in an SGPR by the compiler)
KG_HD u64 mad_u64_k(u32 a, u32 k, u64 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "s"(k));
  return acc;
#else
  return acc + (u64)a * k;
#endif
}

template <int I, int N, typename F>
KG_HD void fe_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    fe_static_for<I + 1, N>(f);
  }
}
// N multiply-adds into one accumulator as ONE asm statement.  The compiler pads every asm statement that defines a
// VGPR with an s_nop before the next reader (it has to assume a dst_sel forwarding hazard), which is one s_nop
// per multiply-add when each is a statement of its own.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KG_NO_MAD_CHAINS)
#define KG_MAD_LINE(x, y) "v_mad_u64_u32 %0, %1, %" #x ", %" #y ", %0\n"
template <int N>
__device__ __forceinline__ u64 mad_chain(const u32 *a, const u32 *b, u64 acc);
template <> __device__ __forceinline__ u64 mad_chain<1>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) : "+v"(acc), "=s"(c) : "v"(a[0]), "v"(b[0])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<2>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) : "+v"(acc), "=s"(c) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<3>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<4>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<5>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<6>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<7>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             KG_MAD_LINE(14, 15) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5]), "v"(a[6]), "v"(b[6])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<8>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             KG_MAD_LINE(14, 15) KG_MAD_LINE(16, 17) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5]), "v"(a[6]), "v"(b[6]), "v"(a[7]), "v"(b[7])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<9>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             KG_MAD_LINE(14, 15) KG_MAD_LINE(16, 17) KG_MAD_LINE(18, 19) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5]), "v"(a[6]), "v"(b[6]), "v"(a[7]), "v"(b[7]), "v"(a[8]), "v"(b[8])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<10>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             KG_MAD_LINE(14, 15) KG_MAD_LINE(16, 17) KG_MAD_LINE(18, 19) KG_MAD_LINE(20, 21) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5]), "v"(a[6]), "v"(b[6]), "v"(a[7]), "v"(b[7]), "v"(a[8]), "v"(b[8]), "v"(a[9]), "v"(b[9])); return acc;
}
template <> __device__ __forceinline__ u64 mad_chain<11>(const u32 *a, const u32 *b, u64 acc) {
  u64 c; asm(KG_MAD_LINE(2, 3) KG_MAD_LINE(4, 5) KG_MAD_LINE(6, 7) KG_MAD_LINE(8, 9) KG_MAD_LINE(10, 11) KG_MAD_LINE(12, 13)
             KG_MAD_LINE(14, 15) KG_MAD_LINE(16, 17) KG_MAD_LINE(18, 19) KG_MAD_LINE(20, 21) KG_MAD_LINE(22, 23) : "+v"(acc), "=s"(c)
             : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]), "v"(a[4]), "v"(b[4]),
               "v"(a[5]), "v"(b[5]), "v"(a[6]), "v"(b[6]), "v"(a[7]), "v"(b[7]), "v"(a[8]), "v"(b[8]), "v"(a[9]), "v"(b[9]),
               "v"(a[10]), "v"(b[10])); return acc;
}
#else
template <int N>
KG_HD u64 mad_chain(const u32 *a, const u32 *b, u64 acc) {
  for (int i = 0; i < N; ++i) acc = mad_u64(a[i], b[i], acc);
  return acc;
}
#endif
// column k of a * b: sum of a[i] * b[k - i] over i in [LO, LO + N)
template <int LO, int N>
KG_HD u64 mad_column(const u32 *a, const u32 *b, int k, u64 acc) {
  u32 ua[N], ub[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    ua[j] = a[LO + j];
    ub[j] = b[k - LO - j];
  }
  return mad_chain<N>(ua, ub, acc);
}

KG_HD void fe_reduce_cols(fe &r, u32 (&t)[9], u64 t9acc) {
  // t: columns 0..8 carried to 29 bits with the fold of the high half applied;
  // t9acc (< 2^36, because wide limbs push up to 2^34 of carry out of column 8): weight 2^261.
  // 2^261 = R0 + R1 * 2^29 (mod p) and 2^256 = 977 + 8 * 2^29 (mod p).
  // Only limbs 0 and 1 are carried; what is left (< 2^16) is parked in limb 2, which the weak
  // form allows, so there is no carry chain through limbs 3..8.
  const u32 t9lo = (u32)t9acc & M29;
  const u32 t9hi = (u32)(t9acc >> 29);  // weight 2^290 = (R0 + R1 * 2^29) * 2^29
  const u32 v8 = t[8] >> 24;
  u64 c = (u64)(t[0] + v8 * 977u);      // < 2^29 + 2^15
  c = mad_u64_k(t9lo, FE_R0, c);
  r.l[0] = (u32)c & M29;
  c >>= 29;
  c += (u64)(t[1] + (v8 << 3) + t9hi * FE_R0);  // < 2^29 + 2^8 + 2^22
  c = mad_u64_k(t9lo, FE_R1, c);
  r.l[1] = (u32)c & M29;
  r.l[2] = t[2] + (u32)(c >> 29) + (t9hi << 8);  // < 2^29 + 2^9 + 2^15
#pragma unroll
  for (int i = 3; i < 8; ++i) r.l[i] = t[i];
  r.l[8] = t[8] & 0xFFFFFFu;
}

KG_HD void fe_mul(fe &r, const fe &a, const fe &b) {
  u32 h[9], t[9];
  u64 acc = 0;
  // columns 9..16 (weight 2^261 and up), carried into 29-bit limbs h[0..8]
  fe_static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    acc = mad_column<k - 8, 17 - k>(a.l, b.l, k, acc);
    h[k - 9] = (u32)acc & M29;
    acc >>= 29;
  });
  h[8] = (u32)acc;
  // columns 0..8 plus h * (2^37 + 0x7A20)
  acc = 0;
  fe_static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int m = k + 1 + (k > 0 ? 2 : 1);  // products of the column + the fold of h[k] (and h[k-1])
    u32 ua[m], ub[m];
#pragma unroll
    for (int j = 0; j <= k; ++j) {
      ua[j] = a.l[j];
      ub[j] = b.l[k - j];
    }
    ua[k + 1] = h[k];
    ub[k + 1] = FE_R0;
    if (k > 0) {
      ua[m - 1] = h[k - 1];
      ub[m - 1] = FE_R1;
    }
    acc = mad_chain<m>(ua, ub, acc);
    t[k] = (u32)acc & M29;
    acc >>= 29;
  });
  acc = mad_u64_k(h[8], FE_R1, acc);
  fe_reduce_cols(r, t, acc);
}

// ---- squaring: a weak -> weak ---------------------------------------------------------
KG_HD void fe_sqr(fe &r, const fe &a) {
  u32 h[9], t[9], d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
  u64 acc = 0;
  // column k of a^2: 2 * a[i] * a[k - i] for 2 i < k, plus a[k/2]^2 when k is even, as one chain
  auto sqr_column = [&](auto kc, u64 c) {
    constexpr int k = decltype(kc)::value;
    constexpr int lo = k > 8 ? k - 8 : 0, hi = (k + 1) / 2, n = hi - lo, m = n + ((k & 1) == 0 ? 1 : 0);
    u32 ua[m], ub[m];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      ua[j] = d[lo + j];
      ub[j] = a.l[k - lo - j];
    }
    if ((k & 1) == 0) ua[m - 1] = ub[m - 1] = a.l[k / 2];
    return mad_chain<m>(ua, ub, c);
  };
  fe_static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    acc = sqr_column(kc, acc);
    h[k - 9] = (u32)acc & M29;
    acc >>= 29;
  });
  h[8] = (u32)acc;
  acc = 0;
  fe_static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int hi = (k + 1) / 2, sq = (k & 1) == 0 ? 1 : 0, m = hi + sq + (k > 0 ? 2 : 1);
    u32 ua[m], ub[m];
#pragma unroll
    for (int j = 0; j < hi; ++j) {
      ua[j] = d[j];
      ub[j] = a.l[k - j];
    }
    if (sq) ua[hi] = ub[hi] = a.l[k / 2];
    ua[hi + sq] = h[k];
    ub[hi + sq] = FE_R0;
    if (k > 0) {
      ua[m - 1] = h[k - 1];
      ub[m - 1] = FE_R1;
    }
    acc = mad_chain<m>(ua, ub, acc);
    t[k] = (u32)acc & M29;
    acc >>= 29;
  });
  acc = mad_u64_k(h[8], FE_R1, acc);
  fe_reduce_cols(r, t, acc);
}

KG_HD bool fe_is_zero_canon(const fe &a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= a.l[i];
  return o == 0;
}

}  // namespace kg
