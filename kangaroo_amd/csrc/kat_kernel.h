// kat_kernel.h — known-answer kernel for the DEVICE build of the field arithmetic (fe29.h, modinv30.h).
//
// The walk kernels exercise fe_mul / fe_sqr / fe_inv on random data only; this kernel runs chosen operands (0, 1,
// p-1, p, values >= p, 2^256-1, the reference's golden vectors) through the very same inline-asm multiplier, one
// operation per lane, so that a test can compare the device results with the oracle.  It plays the role of the
// reference's Int::Check (math/cpu/int.cpp:1158-1511: ModMulK1 / ModSquareK1 / ModInv identities), on the GPU.
#pragma once
#include "fe29.h"
#include "modinv30.h"

namespace kg {

// operations: all operands and results are 8 little-endian 32-bit words (any 256-bit value, also >= p)
enum KatOp : int {
  KAT_MUL = 0,     // a * b
  KAT_SQR = 1,     // a^2
  KAT_SUB = 2,     // a - b          (fe_sub_wide + fe_canon)
  KAT_SUB2 = 3,    // a - b - c      (fe_sub2_wide + fe_canon)
  KAT_SUBMUL = 4,  // (a - b) * c    (wide x weak multiplication)
  KAT_CANON = 5,   // a mod p
  KAT_INV = 6,     // a^-1, 0 for a = 0 (the inversion the walk kernels use)
  KAT_WEAKMUL = 7, // (a * b) * c with the uncanonicalised (weak) product as first operand, then squared: ((ab)c)^2
  KAT_SUBWEAK = 8, // fe_weak(a - b) * c: the carry-and-fold path used for y in the walk
  // raw-limb operations: the nine 29-bit limbs are a[0..7] and b[0] as given (weak ranges, not packed words)
  KAT_SETTLE01 = 9,  // fe_settle<0,1> (what the walk does to y before it is packed into eight words), pack, unpack
  KAT_SETTLE2 = 10,  // fe_settle<2,2> (the stored partial product), pack, unpack
  KAT_NOPS = 11
};

__global__ void __launch_bounds__(256) kg_kat_kernel(int op, const u32 *a, const u32 *b, const u32 *c, u32 *r, u32 n) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u32 wa[8], wb[8], wc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wa[k] = a[8 * i + k];
    wb[k] = b[8 * i + k];
    wc[k] = c[8 * i + k];
  }
  fe x, y, z, t, o;
  fe_from_words(x, wa);
  fe_from_words(y, wb);
  fe_from_words(z, wc);
  switch (op) {
    case KAT_MUL: fe_mul(o, x, y); break;
    case KAT_SQR: fe_sqr(o, x); break;
    case KAT_SUB: fe_sub_wide(o, x, y); break;
    case KAT_SUB2: fe_sub2_wide(o, x, y, z); break;
    case KAT_SUBMUL: fe_sub_wide(t, x, y); fe_mul(o, t, z); break;
    case KAT_CANON: o = x; break;
    case KAT_INV: fe_inv(o, x); break;
    case KAT_WEAKMUL: fe_mul(t, x, y); fe_mul(t, t, z); fe_sqr(o, t); break;
    case KAT_SUBWEAK: fe_sub_wide(t, x, y); fe_weak(t); fe_mul(o, t, z); break;
    case KAT_SETTLE01:
    case KAT_SETTLE2: {
#pragma unroll
      for (int k = 0; k < 8; ++k) t.l[k] = wa[k];
      t.l[8] = wb[0];
      if (op == KAT_SETTLE01) fe_settle<0, 1>(t);
      else fe_settle<2, 2>(t);
      u32 pw[8];
      fe_to_words(pw, t);  // must be exact: every limb below 2^29, limb 8 below 2^24
      fe_from_words(o, pw);
      break;
    }
    default: o = x; break;
  }
  fe_canon(o);
  u32 w[8];
  fe_to_words(w, o);
#pragma unroll
  for (int k = 0; k < 8; ++k) r[8 * i + k] = w[k];
}

}  // namespace kg
