/*
 * kg_oracle.c — CPU restatement of the reference's kangaroo jump path.
 * TEST INFRASTRUCTURE ONLY (see kg_oracle.h).  Plain C11 + unsigned __int128.
 * Citations are to files under /root/reference.
 */
#include "kg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* p = 2^256 - 0x1000003D1 (math/secp256k1.cpp:8) */
static const u64 P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                         0xFFFFFFFFFFFFFFFFULL};
/* group order n (math/secp256k1.cpp:17) */
static const u64 N[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL,
                         0xFFFFFFFFFFFFFFFFULL};
static const u64 GX[4] = {0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL,
                          0x79BE667EF9DCBBACULL};
static const u64 GY[4] = {0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL,
                          0x483ADA7726A3C465ULL};
#define KK 0x1000003D1ULL

/* ---------------------------------------------------------------- helpers */
static int is_zero(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static int cmp256(const u64 a[4], const u64 b[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  }
  return 0;
}
/* r = a + b, returns carry */
static u64 add256(u64 r[4], const u64 a[4], const u64 b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
/* r = a - b, returns borrow */
static u64 sub256(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a[i] - b[i] - br;
    r[i] = (u64)t;
    br = (u64)(t >> 64) & 1;
  }
  return br;
}

/* 512-bit schoolbook product (the reference builds it from four imm_umul rows,
 * math/cpu/intmod.cpp:555-574; any exact 512-bit product is identical). */
static void mul512(u64 r[8], const u64 a[4], const u64 b[4]) {
  memset(r, 0, 8 * sizeof(u64));
  for (int i = 0; i < 4; ++i) {
    u64 c = 0;
    for (int j = 0; j < 4; ++j) {
      u128 t = (u128)a[j] * b[i] + r[i + j] + c;
      r[i + j] = (u64)t;
      c = (u64)(t >> 64);
    }
    r[i + 4] = c;
  }
}

/* The two-step fold of math/cpu/intmod.cpp:576-592:
 *   t[0..4] = hi * 0x1000003D1 ; r512[0..3] += t[0..3] (carry c)
 *   (al,ah) = (t[4]+c) * 0x1000003D1 ; result = r512[0..3] + (ah:al), final carry dropped. */
static void fold512(u64 r[4], const u64 t[8]) {
  u64 s[4];
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)t[4 + i] * KK + t[i];
    s[i] = (u64)c;
    c >>= 64;
  }
  u128 m = (u128)(u64)c * KK; /* (t[4]+c) fits 64 bits: comment at intmod.cpp:584 */
  u128 d = (u128)s[0] + (u64)m;
  r[0] = (u64)d;
  d >>= 64;
  d += (u128)s[1] + (u64)(m >> 64);
  r[1] = (u64)d;
  d >>= 64;
  d += s[2];
  r[2] = (u64)d;
  d >>= 64;
  d += s[3];
  r[3] = (u64)d; /* carry out of bit 256 is dropped: intmod.cpp:591-592 */
}

void kgo_modmul(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[8];
  mul512(t, a, b);
  fold512(r, t);
}

void kgo_modsqr(u64 r[4], const u64 a[4]) {
  /* math/cpu/intmod.cpp:720-799 computes the exact 512-bit square with the
   * doubled-cross-product schedule; the value equals mul512(a,a). */
  u64 t[8];
  mul512(t, a, a);
  fold512(r, t);
}

void kgo_modsub(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4];
  if (sub256(t, a, b)) add256(t, t, P); /* IsNegative() -> Add(&_P) */
  memcpy(r, t, sizeof(t));
}

void kgo_modneg(u64 r[4], const u64 a[4]) {
  u64 t[4];
  sub256(t, P, a); /* Neg(); Add(&_P) on the 256-bit view */
  memcpy(r, t, sizeof(t));
}

/* exact canonical a*b mod p (helper for inversion and curve set-up) */
static void mulmod_canon(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4];
  kgo_modmul(t, a, b);
  /* lazy value is in [0,2^256); canonicalise */
  while (cmp256(t, P) >= 0) sub256(t, t, P);
  memcpy(r, t, sizeof(t));
}

void kgo_modinv(u64 r[4], const u64 a[4]) {
  /* The reference runs a 62-bit delayed-shift binary GCD (intmod.cpp:153-259) and then
   * normalises the result into [0,p) (intmod.cpp:273-291); non-invertible input gives 0
   * (intmod.cpp:177-180, 268-271).  The canonical inverse is unique, so it is restated
   * here as a^(p-2) mod p with exact reduction. */
  u64 base[4], acc[4] = {1, 0, 0, 0};
  memcpy(base, a, sizeof(base));
  while (cmp256(base, P) >= 0) sub256(base, base, P);
  if (is_zero(base)) {
    memset(r, 0, 4 * sizeof(u64));
    return;
  }
  u64 e[4];
  u64 two[4] = {2, 0, 0, 0};
  sub256(e, P, two);
  for (int bit = 0; bit < 256; ++bit) {
    if ((e[bit >> 6] >> (bit & 63)) & 1) mulmod_canon(acc, acc, base);
    mulmod_canon(base, base, base);
  }
  memcpy(r, acc, sizeof(acc));
}

void kgo_group_modinv(u64 (*v)[4], size_t n, u64 (*subp)[4]) {
  /* math/cpu/intgroup.cpp:17-39, same operation order */
  u64 inverse[4], newValue[4];
  if (n == 0) return;
  memcpy(subp[0], v[0], 32);
  for (size_t i = 1; i < n; ++i) kgo_modmul(subp[i], subp[i - 1], v[i]);
  kgo_modinv(inverse, subp[n - 1]);
  for (size_t i = n - 1; i > 0; --i) {
    kgo_modmul(newValue, subp[i - 1], inverse);
    kgo_modmul(inverse, inverse, v[i]);
    memcpy(v[i], newValue, 32);
  }
  memcpy(v[0], inverse, 32);
}

/* ---------------------------------------------------------------- order */
void kgo_order_add(u64 r[4], const u64 a[4], const u64 b[4]) {
  /* Add(a); Sub(_O); if negative Add(_O)  (320-bit signed in the reference) */
  u64 t[4], u[4];
  u64 c = add256(t, a, b);
  u64 br = sub256(u, t, N);
  if (c == 0 && br) memcpy(r, t, 32); /* went negative -> add back */
  else memcpy(r, u, 32);
}
void kgo_order_sub(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4];
  if (sub256(t, a, b)) add256(t, t, N);
  memcpy(r, t, 32);
}
void kgo_order_neg(u64 r[4], const u64 a[4]) {
  u64 t[4];
  sub256(t, N, a); /* Neg(); Add(_O) */
  memcpy(r, t, 32);
}

/* ---------------------------------------------------------------- curve */
static void canon(u64 a[4]) {
  while (cmp256(a, P) >= 0) sub256(a, a, P);
}
static void addmod_canon(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4];
  u64 c = add256(t, a, b);
  if (c || cmp256(t, P) >= 0) sub256(t, t, P);
  memcpy(r, t, 32);
}

void kgo_ec_add(u64 rx[4], u64 ry[4], const u64 x1[4], const u64 y1[4], const u64 x2[4],
                const u64 y2[4]) {
  /* math/secp256k1.cpp:196-216 */
  u64 dy[4], dx[4], s[4], p2[4], t[4], ox[4], oy[4];
  kgo_modsub(dy, y2, y1);
  kgo_modsub(dx, x2, x1);
  kgo_modinv(dx, dx);
  kgo_modmul(s, dy, dx);
  kgo_modsqr(p2, s);
  kgo_modsub(ox, p2, x1);
  kgo_modsub(ox, ox, x2);
  kgo_modsub(t, x2, ox);
  kgo_modmul(t, s, t);
  kgo_modsub(oy, t, y2);
  memcpy(rx, ox, 32);
  memcpy(ry, oy, 32);
}

static void ec_double(u64 rx[4], u64 ry[4], const u64 x[4], const u64 y[4]) {
  /* math/secp256k1.cpp:327-352 (DoubleDirect), canonical results */
  u64 s[4], p2[4], a[4], ox[4], oy[4];
  mulmod_canon(s, x, x);
  addmod_canon(p2, s, s);
  addmod_canon(p2, p2, s);
  addmod_canon(a, y, y);
  kgo_modinv(a, a);
  mulmod_canon(s, p2, a);
  mulmod_canon(p2, s, s);
  addmod_canon(a, x, x);
  kgo_modsub(ox, p2, a);
  canon(ox);
  kgo_modsub(a, x, ox);
  mulmod_canon(p2, a, s);
  kgo_modsub(oy, p2, y);
  canon(oy);
  memcpy(rx, ox, 32);
  memcpy(ry, oy, 32);
}

/* Generator table G * 2^(8i) * (j+1) as in Secp256K1::Init (math/secp256k1.cpp:21-31),
 * built lazily; 32 x 255 affine points. */
static u64 (*g_tab_x)[4];
static u64 (*g_tab_y)[4];
static void build_gtable(void) {
  if (g_tab_x) return;
  u64(*tx)[4] = malloc(32 * 256 * 32);
  u64(*ty)[4] = malloc(32 * 256 * 32);
  u64 nx[4], ny[4];
  memcpy(nx, GX, 32);
  memcpy(ny, GY, 32);
  for (int i = 0; i < 32; ++i) {
    memcpy(tx[i * 256], nx, 32);
    memcpy(ty[i * 256], ny, 32);
    ec_double(nx, ny, nx, ny);
    for (int j = 1; j < 255; ++j) {
      memcpy(tx[i * 256 + j], nx, 32);
      memcpy(ty[i * 256 + j], ny, 32);
      kgo_ec_add(nx, ny, nx, ny, tx[i * 256], ty[i * 256]);
      canon(nx);
      canon(ny);
    }
    memcpy(tx[i * 256 + 255], nx, 32);
    memcpy(ty[i * 256 + 255], ny, 32);
  }
  g_tab_y = ty;
  g_tab_x = tx;
}

/* Jacobian-free accumulation: the reference adds table points in projective form
 * (Add2, math/secp256k1.cpp:263-299) and divides by z at the end; the affine result
 * is the same point.  Here: projective (X:Y:Z) mixed addition with exact reduction. */
typedef struct {
  u64 x[4], y[4], z[4];
  int inf;
} jpt;

static void submod_canon(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[4];
  if (sub256(t, a, b)) add256(t, t, P);
  memcpy(r, t, 32);
}

/* homogeneous projective add of affine (x2,y2) into (X1:Y1:Z1): same formulas as Add2 */
static void padd_mixed(jpt *p1, const u64 x2[4], const u64 y2[4]) {
  if (p1->inf) {
    memcpy(p1->x, x2, 32);
    memcpy(p1->y, y2, 32);
    memset(p1->z, 0, 32);
    p1->z[0] = 1;
    p1->inf = 0;
    return;
  }
  u64 u[4], v[4], u1[4], v1[4], us2[4], vs2[4], vs3[4], us2w[4], vs2v2[4], vs3u2[4], a[4], t2[4];
  mulmod_canon(u1, y2, p1->z);
  mulmod_canon(v1, x2, p1->z);
  submod_canon(u, u1, p1->y);
  submod_canon(v, v1, p1->x);
  mulmod_canon(us2, u, u);
  mulmod_canon(vs2, v, v);
  mulmod_canon(vs3, vs2, v);
  mulmod_canon(us2w, us2, p1->z);
  mulmod_canon(vs2v2, vs2, p1->x);
  addmod_canon(t2, vs2v2, vs2v2);
  submod_canon(a, us2w, vs3);
  submod_canon(a, a, t2);
  u64 rx[4], ry[4], rz[4];
  mulmod_canon(rx, v, a);
  mulmod_canon(vs3u2, vs3, p1->y);
  submod_canon(ry, vs2v2, a);
  mulmod_canon(ry, ry, u);
  submod_canon(ry, ry, vs3u2);
  mulmod_canon(rz, vs3, p1->z);
  memcpy(p1->x, rx, 32);
  memcpy(p1->y, ry, 32);
  memcpy(p1->z, rz, 32);
}

static void mul_g_proj(jpt *q, const u64 k[4]) {
  build_gtable();
  q->inf = 1;
  for (int i = 0; i < 32; ++i) {
    unsigned b = (unsigned)((k[i >> 3] >> ((i & 7) * 8)) & 0xFF);
    if (b) padd_mixed(q, g_tab_x[256 * i + (b - 1)], g_tab_y[256 * i + (b - 1)]);
  }
}

int kgo_ec_mul_g(u64 x[4], u64 y[4], const u64 k[4]) {
  jpt q;
  mul_g_proj(&q, k);
  if (q.inf) {
    memset(x, 0, 32);
    memset(y, 0, 32);
    return 0;
  }
  u64 zi[4];
  kgo_modinv(zi, q.z);
  mulmod_canon(x, q.x, zi);
  mulmod_canon(y, q.y, zi);
  return 1;
}

void kgo_ec_mul_g_batch(u64 (*x)[4], u64 (*y)[4], const u64 (*k)[4], size_t n) {
  /* math/secp256k1.cpp:60-82: projective results, one grouped inversion of all z */
  if (n == 0) return;
  jpt *q = malloc(n * sizeof(jpt));
  u64(*z)[4] = malloc(n * 32);
  u64(*scr)[4] = malloc(n * 32);
  for (size_t i = 0; i < n; ++i) {
    mul_g_proj(&q[i], k[i]);
    if (q[i].inf) {
      memset(z[i], 0, 32);
      z[i][0] = 1;
    } else
      memcpy(z[i], q[i].z, 32);
  }
  kgo_group_modinv(z, n, scr);
  for (size_t i = 0; i < n; ++i) {
    if (q[i].inf) {
      memset(x[i], 0, 32);
      memset(y[i], 0, 32);
    } else {
      mulmod_canon(x[i], q[i].x, z[i]);
      mulmod_canon(y[i], q[i].y, z[i]);
    }
  }
  free(q);
  free(z);
  free(scr);
}

int kgo_ec_decompress(u64 y[4], const u64 x[4], int y_is_odd) {
  /* y^2 = x^3 + 7; p = 3 mod 4 so sqrt = v^((p+1)/4)  (math/cpu/intmod.cpp:349-354) */
  u64 v[4], s[4], seven[4] = {7, 0, 0, 0};
  mulmod_canon(s, x, x);
  mulmod_canon(v, s, x);
  addmod_canon(v, v, seven);
  u64 e[4], one[4] = {1, 0, 0, 0};
  add256(e, P, one); /* p+1 overflows 256 bits? p+1 = 2^256 - 0x1000003D0 fits */
  /* e = (p+1)/4 */
  for (int sft = 0; sft < 2; ++sft) {
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? (e[i + 1] << 63) : 0);
  }
  u64 acc[4] = {1, 0, 0, 0}, base[4];
  memcpy(base, v, 32);
  for (int bit = 0; bit < 256; ++bit) {
    if ((e[bit >> 6] >> (bit & 63)) & 1) mulmod_canon(acc, acc, base);
    mulmod_canon(base, base, base);
  }
  mulmod_canon(s, acc, acc);
  if (cmp256(s, v) != 0) return 0;
  if ((int)(acc[0] & 1) != (y_is_odd ? 1 : 0)) kgo_modneg(acc, acc);
  memcpy(y, acc, 32);
  return 1;
}

/* ---------------------------------------------------------------- RNG */
void kgo_rseed(kgo_rng *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
uint32_t kgo_rndl(kgo_rng *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t yv = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (yv >> 1);
      if (yv & 1) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t yv = g->mt[g->idx++];
  yv ^= yv >> 11;
  yv ^= (yv << 7) & 0x9d2c5680u;
  yv ^= (yv << 15) & 0xefc60000u;
  yv ^= yv >> 18;
  return yv;
}
void kgo_rand_bits(kgo_rng *g, u64 r[4], int nbit) {
  /* math/cpu/int.cpp:717-730: note the extra masked draw even when nbit%32==0 */
  uint32_t w[10];
  memset(w, 0, sizeof(w));
  uint32_t nb = (uint32_t)nbit / 32, left = (uint32_t)nbit % 32;
  uint32_t mask = (1u << left) - 1u;
  uint32_t i = 0;
  for (; i < nb; ++i) w[i] = kgo_rndl(g);
  w[i] = kgo_rndl(g) & mask;
  for (int k = 0; k < 4; ++k) r[k] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
}

/* ---------------------------------------------------------------- set-up */
static double to_double(const u64 a[4]) {
  /* Int::ToDouble: ascending sum of 32-bit words times 2^(32 i)  (math/cpu/int.cpp:625-637) */
  double base = 1.0, sum = 0.0;
  const double pw32 = 4294967296.0;
  for (int i = 0; i < 8; ++i) {
    uint32_t w = (uint32_t)(a[i >> 1] >> ((i & 1) * 32));
    sum += (double)w * base;
    base *= pw32;
  }
  return sum;
}

int kgo_create_jump_table(int rangePower, u64 jD[KGO_NB_JUMP][4], u64 jPx[KGO_NB_JUMP][4],
                          u64 jPy[KGO_NB_JUMP][4]) {
  /* core/kangaroo.cpp:575-619 */
  int jumpBit = rangePower / 2 + 1;
  int maxRetry = 100, tries = 0, ok = 0;
  double maxAvg = pow(2.0, (double)jumpBit - 0.95);
  double minAvg = pow(2.0, (double)jumpBit - 1.05);
  kgo_rng g;
  kgo_rseed(&g, 0x600DCAFEu);
  while (!ok && maxRetry > 0) {
    u64 total[4] = {0, 0, 0, 0};
    for (int i = 0; i < KGO_NB_JUMP; ++i) {
      kgo_rand_bits(&g, jD[i], jumpBit);
      if (is_zero(jD[i])) jD[i][0] = 1;
      add256(total, total, jD[i]);
    }
    /* Int::ToDouble sums limbs as doubles (math/cpu/int.cpp) */
    double avg = to_double(total) / (double)KGO_NB_JUMP;
    ok = avg > minAvg && avg < maxAvg;
    maxRetry--;
    tries++;
  }
  kgo_ec_mul_g_batch(jPx, jPy, (const u64(*)[4])jD, KGO_NB_JUMP);
  return tries;
}

void kgo_create_herd(kgo_rng *g, int rangePower, const u64 rangeWidthDiv8[4], const u64 keyX[4],
                     const u64 keyY[4], int firstType, size_t n, u64 (*px)[4], u64 (*py)[4],
                     u64 (*d)[4]) {
  /* core/kangaroo.cpp:521-573 */
  int offset = firstType % 2;
  for (size_t j = 0; j < n; ++j) {
    if (((j + (size_t)offset) % 2) == 0) {
      kgo_rand_bits(g, d[j], rangePower);
    } else {
      kgo_rand_bits(g, d[j], rangePower - 2);
      kgo_order_sub(d[j], d[j], rangeWidthDiv8);
    }
  }
  kgo_ec_mul_g_batch(px, py, (const u64(*)[4])d, n);
  /* wild: S = keyToSearch + S with one grouped inversion over the wild subset
   * (vector AddDirect, math/secp256k1.cpp:217-261: p1 = key, p2 = S) */
  for (size_t j = 0; j < n; ++j) {
    if (((j + (size_t)offset) % 2) != 0) {
      if (is_zero(keyX)) continue; /* p1.x == 0 -> result is p2 (secp256k1.cpp:240-241) */
      u64 rx[4], ry[4];
      kgo_ec_add(rx, ry, keyX, keyY, px[j], py[j]);
      memcpy(px[j], rx, 32);
      memcpy(py[j], ry, 32);
    }
  }
}

/* ---------------------------------------------------------------- walk */
size_t kgo_walk(u64 (*px)[4], u64 (*py)[4], u64 (*d)[4], size_t n, u64 nsteps, u64 dpMask,
                const u64 jD[KGO_NB_JUMP][4], const u64 jPx[KGO_NB_JUMP][4],
                const u64 jPy[KGO_NB_JUMP][4], kgo_dp *out, size_t cap) {
  size_t found = 0;
  u64(*dx)[4] = malloc(n * 32);
  u64(*scr)[4] = malloc(n * 32);
  unsigned *jmp = malloc(n * sizeof(unsigned));
  for (u64 step = 1; step <= nsteps; ++step) {
    /* first pass: core/kangaroo.cpp:290-299 */
    for (size_t g = 0; g < n; ++g) {
      jmp[g] = (unsigned)(px[g][0] & (KGO_NB_JUMP - 1));
      kgo_modsub(dx[g], px[g], jPx[jmp[g]]);
    }
    kgo_group_modinv(dx, n, scr); /* core/kangaroo.cpp:301-302 */
    /* second pass: core/kangaroo.cpp:305-325 */
    for (size_t g = 0; g < n; ++g) {
      u64 dy[4], s[4], p2[4], rx[4], ry[4];
      unsigned j = jmp[g];
      kgo_modsub(dy, py[g], jPy[j]);
      kgo_modmul(s, dy, dx[g]);
      kgo_modsqr(p2, s);
      kgo_modsub(rx, p2, jPx[j]);
      kgo_modsub(rx, rx, px[g]);
      kgo_modsub(ry, px[g], rx);
      kgo_modmul(ry, ry, s);
      kgo_modsub(ry, ry, py[g]);
      memcpy(px[g], rx, 32);
      memcpy(py[g], ry, 32);
      kgo_order_add(d[g], d[g], jD[j]);
      if ((px[g][3] & dpMask) == 0) { /* IsDP: core/kangaroo.cpp:104-108 */
        if (found < cap) {
          memcpy(out[found].x, px[g], 32);
          memcpy(out[found].d, d[g], 32);
          out[found].kIdx = g;
          out[found].step = step;
        }
        found++;
      }
    }
  }
  free(dx);
  free(scr);
  free(jmp);
  return found;
}

void kgo_ht_convert(const u64 x[4], const u64 d[4], uint32_t type, u64 *h, u64 X[4], u64 D[4]) {
  /* memory/hashtable.cpp:65-89 */
  u64 sign = 0;
  u64 type64 = (u64)type << 62;
  memcpy(X, x, 32);
  if (d[3] > 0x7FFFFFFFFFFFFFFFULL) {
    u64 nn[4];
    kgo_order_neg(nn, d);
    D[0] = nn[0];
    D[1] = nn[1];
    D[2] = nn[2];
    D[3] = nn[3] & 0x3FFFFFFFFFFFFFFFULL;
    sign = 1ULL << 63;
  } else {
    D[0] = d[0];
    D[1] = d[1];
    D[2] = d[2];
    D[3] = d[3] & 0x3FFFFFFFFFFFFFFFULL;
  }
  D[1] |= sign;
  D[1] |= type64;
  *h = x[2] & 0x3FFFF; /* HASH_MASK, constants.h:33-35 */
}
