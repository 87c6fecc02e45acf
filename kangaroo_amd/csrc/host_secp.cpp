// host_secp.cpp — host-side secp256k1 set-up and DP bookkeeping (include/kangaroo_amd_host.h).
//
// Product code (not the oracle): what the reference does on the CPU around the GPU engine —
// jump-table and herd creation (core/kangaroo.cpp:521-626), public-key parsing
// (math/secp256k1.cpp:109-168), the DP table and collision resolution
// (memory/hashtable.cpp:65-226, core/kangaroo.cpp:155-256) — written from scratch on
// 4x64-bit limbs with unsigned __int128, always fully reduced.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <random>
#include <unordered_map>
#include <vector>

#include "../../include/kangaroo_amd.h"
#include "../../include/kangaroo_amd_host.h"
#include "fe29.h"
#include "modinv30.h"

namespace {

typedef unsigned __int128 u128;
typedef uint64_t u64;

struct U256 {
  u64 v[4];
};

const U256 FP = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
const U256 ORD = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
const U256 G_X = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
const U256 G_Y = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};
const u64 PK = 0x1000003D1ULL;

inline U256 load(const u64 *p) {
  U256 r;
  memcpy(r.v, p, 32);
  return r;
}
inline void store(u64 *p, const U256 &a) { memcpy(p, a.v, 32); }
inline bool is_zero(const U256 &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline int cmp(const U256 &a, const U256 &b) {
  for (int i = 3; i >= 0; --i)
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i] ? 1 : -1;
  return 0;
}
inline u64 add_to(U256 &r, const U256 &a, const U256 &b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
inline u64 sub_to(U256 &r, const U256 &a, const U256 &b) {
  u64 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a.v[i] - b.v[i] - br;
    r.v[i] = (u64)t;
    br = (u64)(t >> 64) & 1;
  }
  return br;
}
inline int bit_length(const U256 &a) {
  for (int i = 3; i >= 0; --i)
    if (a.v[i]) return 64 * i + 64 - __builtin_clzll(a.v[i]);
  return 0;
}
inline U256 shr(const U256 &a, int n) {  // n < 64
  U256 r;
  for (int i = 0; i < 4; ++i) r.v[i] = (a.v[i] >> n) | ((i < 3 && n) ? (a.v[i + 1] << (64 - n)) : 0);
  return r;
}

// ---- generic modular add/sub (m = FP or ORD), operands < m -------------------------------------
inline U256 mod_add(const U256 &a, const U256 &b, const U256 &m) {
  U256 t, s;
  u64 c = add_to(t, a, b);
  u64 br = sub_to(s, t, m);
  return (c || !br) ? s : t;
}
inline U256 mod_sub(const U256 &a, const U256 &b, const U256 &m) {
  U256 t;
  if (sub_to(t, a, b)) add_to(t, t, m);
  return t;
}
inline U256 mod_neg(const U256 &a, const U256 &m) {
  if (is_zero(a)) return a;
  U256 t;
  sub_to(t, m, a);
  return t;
}
inline U256 reduce_once(const U256 &a, const U256 &m) {
  U256 s;
  return sub_to(s, a, m) ? a : s;
}

// ---- field multiplication, canonical result ----------------------------------------------------------
U256 fmul(const U256 &a, const U256 &b) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 c = 0;
    for (int j = 0; j < 4; ++j) {
      u128 w = (u128)a.v[j] * b.v[i] + t[i + j] + c;
      t[i + j] = (u64)w;
      c = (u64)(w >> 64);
    }
    t[i + 4] = c;
  }
  // fold 2^256 = PK twice, then bring into [0,p)
  u128 c = 0;
  U256 r;
  for (int i = 0; i < 4; ++i) {
    c += (u128)t[4 + i] * PK + t[i];
    r.v[i] = (u64)c;
    c >>= 64;
  }
  u128 m = (u128)(u64)c * PK;
  c = (u128)r.v[0] + (u64)m;
  r.v[0] = (u64)c;
  c >>= 64;
  c += (u128)r.v[1] + (u64)(m >> 64);
  r.v[1] = (u64)c;
  c >>= 64;
  c += r.v[2];
  r.v[2] = (u64)c;
  c >>= 64;
  c += r.v[3];
  r.v[3] = (u64)c;
  c >>= 64;
  if ((u64)c) {  // one more 2^256
    U256 k = {{PK, 0, 0, 0}};
    add_to(r, r, k);
  }
  return reduce_once(r, FP);
}
inline U256 fsqr(const U256 &a) { return fmul(a, a); }
inline U256 fadd(const U256 &a, const U256 &b) { return mod_add(a, b, FP); }
inline U256 fsub(const U256 &a, const U256 &b) { return mod_sub(a, b, FP); }

// inversion through the engine's own constant-time safegcd (modinv30.h)
U256 finv(const U256 &a) {
  uint32_t w[8];
  for (int i = 0; i < 4; ++i) {
    w[2 * i] = (uint32_t)a.v[i];
    w[2 * i + 1] = (uint32_t)(a.v[i] >> 32);
  }
  kg::fe f, r;
  kg::fe_from_words(f, w);
  kg::fe_inv(r, f);
  kg::fe_canon(r);
  kg::fe_to_words(w, r);
  U256 o;
  for (int i = 0; i < 4; ++i) o.v[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
  return o;
}

// in-place batch inversion (Montgomery), zero entries are left as zero
void batch_inv(std::vector<U256> &v) {
  const size_t n = v.size();
  std::vector<U256> pre(n);
  U256 acc = {{1, 0, 0, 0}};
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    if (!is_zero(v[i])) acc = fmul(acc, v[i]);
  }
  U256 inv = finv(acc);
  for (size_t i = n; i-- > 0;) {
    if (is_zero(v[i])) continue;
    U256 t = fmul(inv, pre[i]);
    inv = fmul(inv, v[i]);
    v[i] = t;
  }
}

// ---- points -------------------------------------------------------------------------------------------------
struct Aff {
  U256 x, y;
  bool inf;
};
struct Jac {
  U256 x, y, z;
  bool inf;
};

Aff aff_add(const Aff &p, const Aff &q);
Aff aff_dbl(const Aff &p) {
  if (p.inf || is_zero(p.y)) return Aff{{}, {}, true};
  U256 xx = fsqr(p.x);
  U256 num = fadd(fadd(xx, xx), xx);
  U256 s = fmul(num, finv(fadd(p.y, p.y)));
  U256 rx = fsub(fsub(fsqr(s), p.x), p.x);
  U256 ry = fsub(fmul(s, fsub(p.x, rx)), p.y);
  return Aff{rx, ry, false};
}
Aff aff_add(const Aff &p, const Aff &q) {
  if (p.inf) return q;
  if (q.inf) return p;
  if (cmp(p.x, q.x) == 0) {
    if (cmp(p.y, q.y) == 0) return aff_dbl(p);
    return Aff{{}, {}, true};
  }
  U256 s = fmul(fsub(q.y, p.y), finv(fsub(q.x, p.x)));
  U256 rx = fsub(fsub(fsqr(s), p.x), q.x);
  U256 ry = fsub(fmul(s, fsub(p.x, rx)), p.y);
  return Aff{rx, ry, false};
}

// Jacobian doubling / mixed addition (a = 0 curve)
Jac jac_dbl(const Jac &p) {
  if (p.inf || is_zero(p.y)) return Jac{{}, {}, {}, true};
  U256 a = fsqr(p.x), b = fsqr(p.y), c = fsqr(b);
  U256 t = fadd(p.x, b);
  U256 d = fsub(fsub(fsqr(t), a), c);
  d = fadd(d, d);
  U256 e = fadd(fadd(a, a), a);
  U256 f = fsqr(e);
  Jac r;
  r.inf = false;
  r.x = fsub(f, fadd(d, d));
  U256 c8 = fadd(c, c);
  c8 = fadd(c8, c8);
  c8 = fadd(c8, c8);
  r.y = fsub(fmul(e, fsub(d, r.x)), c8);
  U256 yz = fmul(p.y, p.z);
  r.z = fadd(yz, yz);
  return r;
}
Jac jac_add_mixed(const Jac &p, const Aff &q) {
  if (q.inf) return p;
  if (p.inf) return Jac{q.x, q.y, {{1, 0, 0, 0}}, false};
  U256 z2 = fsqr(p.z);
  U256 u2 = fmul(q.x, z2);
  U256 s2 = fmul(q.y, fmul(z2, p.z));
  U256 h = fsub(u2, p.x);
  U256 rr = fsub(s2, p.y);
  if (is_zero(h)) {
    if (is_zero(rr)) return jac_dbl(p);
    return Jac{{}, {}, {}, true};
  }
  U256 h2 = fsqr(h), h3 = fmul(h2, h), v = fmul(p.x, h2);
  Jac r;
  r.inf = false;
  r.x = fsub(fsub(fsqr(rr), h3), fadd(v, v));
  r.y = fsub(fmul(rr, fsub(v, r.x)), fmul(p.y, h3));
  r.z = fmul(p.z, h);
  return r;
}

// fixed-base table: T[i][j] = (j+1) * 256^i * G, i < 32, j < 255 (same shape as Secp256K1::Init)
std::vector<Aff> g_table;
std::once_flag g_table_once;
void build_table() {
  g_table.resize(32 * 255);
  Aff base{G_X, G_Y, false};
  for (int i = 0; i < 32; ++i) {
    Aff cur = base;
    for (int j = 0; j < 255; ++j) {
      g_table[i * 255 + j] = cur;
      cur = aff_add(cur, base);
    }
    base = cur;  // 256 * base
  }
}

Jac mul_g_jac(const U256 &k) {
  std::call_once(g_table_once, build_table);
  Jac acc{{}, {}, {}, true};
  for (int i = 0; i < 32; ++i) {
    unsigned b = (unsigned)((k.v[i >> 3] >> ((i & 7) * 8)) & 0xFF);
    if (b) acc = jac_add_mixed(acc, g_table[i * 255 + (b - 1)]);
  }
  return acc;
}
Aff to_affine(const Jac &p) {
  if (p.inf) return Aff{{}, {}, true};
  U256 zi = finv(p.z), zi2 = fsqr(zi);
  return Aff{fmul(p.x, zi2), fmul(p.y, fmul(zi2, zi)), false};
}
Aff mul_g(const U256 &k) { return to_affine(mul_g_jac(k)); }

void mul_g_batch(const u64 *k, size_t n, u64 *x, u64 *y) {
  std::vector<Jac> pts(n);
  std::vector<U256> zs(n);
  for (size_t i = 0; i < n; ++i) {
    pts[i] = mul_g_jac(load(k + 4 * i));
    zs[i] = pts[i].inf ? U256{{0, 0, 0, 0}} : pts[i].z;
  }
  batch_inv(zs);
  for (size_t i = 0; i < n; ++i) {
    U256 ox = {{0, 0, 0, 0}}, oy = {{0, 0, 0, 0}};
    if (!pts[i].inf) {
      U256 zi2 = fsqr(zs[i]);
      ox = fmul(pts[i].x, zi2);
      oy = fmul(pts[i].y, fmul(zi2, zs[i]));
    }
    store(x + 4 * i, ox);
    store(y + 4 * i, oy);
  }
}

// y from x: y^2 = x^3 + 7, p = 3 mod 4
bool lift_x(const U256 &x, bool odd, U256 &y) {
  U256 seven = {{7, 0, 0, 0}};
  U256 rhs = fadd(fmul(fsqr(x), x), seven);
  // exponent (p+1)/4
  U256 e;
  U256 one = {{1, 0, 0, 0}};
  add_to(e, FP, one);
  e = shr(e, 2);
  U256 acc = one, base = rhs;
  for (int bit = 0; bit < 256; ++bit) {
    if ((e.v[bit >> 6] >> (bit & 63)) & 1) acc = fmul(acc, base);
    base = fsqr(base);
  }
  if (cmp(fsqr(acc), rhs) != 0) return false;
  if (((acc.v[0] & 1) != 0) != odd) acc = mod_neg(acc, FP);
  y = acc;
  return true;
}

// Int::Rand(nbit): nbit/32 words then one masked word (math/cpu/int.cpp:717-730)
U256 rand_bits(std::mt19937 &g, int nbit) {
  uint32_t w[10] = {0};
  uint32_t nb = (uint32_t)nbit / 32, left = (uint32_t)nbit % 32, i = 0;
  for (; i < nb; ++i) w[i] = (uint32_t)g();
  w[i] = (uint32_t)g() & ((1u << left) - 1u);
  U256 r;
  for (int k = 0; k < 4; ++k) r.v[k] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
  return r;
}

}  // namespace

// =================================================================================================
extern "C" {

int kgh_ec_mul_g(uint64_t x[4], uint64_t y[4], const uint64_t k[4]) {
  Aff p = mul_g(load(k));
  if (p.inf) {
    memset(x, 0, 32);
    memset(y, 0, 32);
    return 0;
  }
  store(x, p.x);
  store(y, p.y);
  return 1;
}
int kgh_ec_mul_g_batch(uint64_t *x, uint64_t *y, const uint64_t *k, size_t n) {
  mul_g_batch(k, n, x, y);
  return 0;
}
int kgh_ec_add(uint64_t rx[4], uint64_t ry[4], const uint64_t x1[4], const uint64_t y1[4], const uint64_t x2[4],
               const uint64_t y2[4]) {
  Aff r = aff_add(Aff{load(x1), load(y1), false}, Aff{load(x2), load(y2), false});
  if (r.inf) return -1;
  store(rx, r.x);
  store(ry, r.y);
  return 0;
}

int kgh_parse_pubkey(const char *hex, uint64_t x[4], uint64_t y[4]) {
  if (!hex) return -1;
  size_t len = strlen(hex);
  auto hv = [](char c) -> int {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return 10 + c - 'a';
    if (c >= 'A' && c <= 'F') return 10 + c - 'A';
    return -1;
  };
  if (len != 66 && len != 130) return -1;
  std::vector<uint8_t> b(len / 2);
  for (size_t i = 0; i < len / 2; ++i) {
    int h = hv(hex[2 * i]), l = hv(hex[2 * i + 1]);
    if (h < 0 || l < 0) return -1;
    b[i] = (uint8_t)(h * 16 + l);
  }
  auto be = [&](size_t off) {
    U256 r = {{0, 0, 0, 0}};
    for (int i = 0; i < 32; ++i) r.v[3 - i / 8] |= (u64)b[off + i] << (8 * (7 - i % 8));
    return r;
  };
  U256 px = be(1), py;
  if ((b[0] == 2 || b[0] == 3) && len == 66) {
    if (cmp(px, FP) >= 0 || !lift_x(px, b[0] == 3, py)) return -2;
  } else if (b[0] == 4 && len == 130) {
    py = be(33);
    U256 seven = {{7, 0, 0, 0}};
    if (cmp(px, FP) >= 0 || cmp(py, FP) >= 0 || cmp(fsqr(py), fadd(fmul(fsqr(px), px), seven)) != 0) return -2;
  } else {
    return -1;
  }
  store(x, px);
  store(y, py);
  return 0;
}

int kgh_create_jump_table(int range_power, uint64_t *jd, uint64_t *jpx, uint64_t *jpy) {
  const int jump_bit = range_power / 2 + 1;
  const double max_avg = std::pow(2.0, (double)jump_bit - 0.95), min_avg = std::pow(2.0, (double)jump_bit - 1.05);
  std::mt19937 g(0x600DCAFEu);
  int tries = 0;
  bool ok = false;
  while (!ok && tries < 100) {
    double sum = 0.0;
    // the reference sums the 64 distances exactly and converts once (Int::ToDouble)
    U256 total = {{0, 0, 0, 0}};
    for (int i = 0; i < KG_NB_JUMP; ++i) {
      U256 d = rand_bits(g, jump_bit);
      if (is_zero(d)) d.v[0] = 1;
      store(jd + 4 * i, d);
      add_to(total, total, d);
    }
    double base = 1.0;
    for (int i = 0; i < 8; ++i) {
      sum += (double)(uint32_t)(total.v[i >> 1] >> ((i & 1) * 32)) * base;
      base *= 4294967296.0;
    }
    const double avg = sum / (double)KG_NB_JUMP;
    ok = avg > min_avg && avg < max_avg;
    ++tries;
  }
  mul_g_batch(jd, KG_NB_JUMP, jpx, jpy);
  return tries;
}

struct kgh_rng {
  std::mt19937 g;
};
kgh_rng *kgh_rng_new(uint32_t seed) {
  kgh_rng *r = new kgh_rng();
  r->g.seed(seed);
  return r;
}
void kgh_rng_free(kgh_rng *r) { delete r; }
uint32_t kgh_rng_next(kgh_rng *r) { return (uint32_t)r->g(); }

int kgh_herd_distances(kgh_rng *r, int range_power, const uint64_t range_width[4], int first_type, size_t n,
                       uint64_t *d) {
  if (!r || !d) return -1;
  const U256 w8 = shr(load(range_width), 3);
  const int off = first_type % 2;
  for (size_t j = 0; j < n; ++j) {
    U256 v;
    if (((j + (size_t)off) % 2) == 0) {
      v = rand_bits(r->g, range_power);
      // reference leaves the raw Rand() value (may exceed n only for 256-bit ranges)
    } else {
      v = rand_bits(r->g, range_power - 2);
      v = mod_sub(v, w8, ORD);
    }
    store(d + 4 * j, v);
  }
  return 0;
}

int kgh_create_herd(kgh_rng *r, int range_power, const uint64_t range_width[4], const uint64_t key_x[4],
                    const uint64_t key_y[4], int first_type, size_t n, uint64_t *px, uint64_t *py, uint64_t *d) {
  if (!r || !px || !py || !d) return -1;
  const int off = first_type % 2;
  kgh_herd_distances(r, range_power, range_width, first_type, n, d);
  mul_g_batch(d, n, px, py);
  // wild kangaroos start at key + d*G: one batched inversion over the wild subset
  const Aff key{load(key_x), load(key_y), false};
  if (is_zero(key.x)) return 0;
  std::vector<size_t> idx;
  std::vector<U256> dx;
  for (size_t j = 0; j < n; ++j)
    if (((j + (size_t)off) % 2) != 0) {
      idx.push_back(j);
      dx.push_back(fsub(load(px + 4 * j), key.x));
    }
  batch_inv(dx);
  for (size_t t = 0; t < idx.size(); ++t) {
    const size_t j = idx[t];
    const U256 x2 = load(px + 4 * j), y2 = load(py + 4 * j);
    const U256 s = fmul(fsub(y2, key.y), dx[t]);
    const U256 rx = fsub(fsub(fsqr(s), key.x), x2);
    const U256 ry = fsub(fmul(s, fsub(x2, rx)), y2);
    store(px + 4 * j, rx);
    store(py + 4 * j, ry);
  }
  return 0;
}

// Fixed-base comb for the GPU herd kernel (kangaroo_amd/csrc/herd_kernel.h): 32 windows of 8 bits.
// Entry (w, j) = (j * 2^(8w) + c_w) * G with fixed offsets c_w, and the start point S = -(sum c_w) * G, so
// that S + sum_w T[w][digit_w(k)] = k * G with an addition in EVERY window (no "digit 0: skip" case and no
// point at infinity on the device).  Points as 9+9 limbs of 29 bits (fe29.h).
int kgh_comb_table(uint32_t *tab, uint64_t start_x[4], uint64_t start_y[4]) {
  if (!tab || !start_x || !start_y) return -1;
  const size_t cnt = 32 * 256;
  std::vector<u64> k(4 * cnt), x(4 * cnt), y(4 * cnt);
  U256 sum = {{0, 0, 0, 0}};
  u64 sm = 0x4B616E6761726F6FULL;  // splitmix64 stream for the offsets
  auto next = [&]() {
    sm += 0x9E3779B97F4A7C15ULL;
    u64 z = sm;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  };
  for (int w = 0; w < 32; ++w) {
    U256 c = {{next(), next(), next(), next() >> 1}};  // < 2^255 < n
    sum = mod_add(sum, c, ORD);
    for (int j = 0; j < 256; ++j) {
      U256 v = {{0, 0, 0, 0}};
      v.v[w / 8] = (u64)j << (8 * (w % 8));
      v = reduce_once(v, ORD);
      store(&k[4 * (size_t)(w * 256 + j)], mod_add(v, c, ORD));
    }
  }
  mul_g_batch(k.data(), cnt, x.data(), y.data());
  for (size_t e = 0; e < cnt; ++e) {
    for (int c = 0; c < 2; ++c) {
      const u64 *src = (c == 0 ? x.data() : y.data()) + 4 * e;
      uint32_t w8[8];
      for (int i = 0; i < 4; ++i) {
        w8[2 * i] = (uint32_t)src[i];
        w8[2 * i + 1] = (uint32_t)(src[i] >> 32);
      }
      kg::fe f;
      kg::fe_from_words(f, w8);
      for (int i = 0; i < 9; ++i) tab[e * 18 + 9 * c + i] = f.l[i];
    }
  }
  const Aff S = mul_g(mod_neg(sum, ORD));
  store(start_x, S.x);
  store(start_y, S.y);
  return 0;
}

int kgh_synthetic_herd(const uint64_t d0[4], const uint64_t stride[4], size_t n, uint64_t *px, uint64_t *py,
                       uint64_t *d) {
  if (!n) return 0;
  const size_t chains = n < 4096 ? n : 4096;
  const U256 st = reduce_once(load(stride), ORD);
  // heads: d0 + c*stride
  std::vector<U256> hd(chains);
  U256 cur = reduce_once(load(d0), ORD);
  for (size_t c = 0; c < chains; ++c) {
    hd[c] = cur;
    cur = mod_add(cur, st, ORD);
  }
  // row step = chains * stride
  U256 row = {{0, 0, 0, 0}};
  for (size_t c = 0; c < chains; ++c) row = mod_add(row, st, ORD);
  const Aff S = mul_g(row);
  std::vector<u64> hk(4 * chains), hx(4 * chains), hy(4 * chains);
  for (size_t c = 0; c < chains; ++c) store(&hk[4 * c], hd[c]);
  mul_g_batch(hk.data(), chains, hx.data(), hy.data());
  std::vector<U256> X(chains), Y(chains), dx(chains);
  for (size_t c = 0; c < chains; ++c) {
    X[c] = load(&hx[4 * c]);
    Y[c] = load(&hy[4 * c]);
  }
  for (size_t base = 0; base < n; base += chains) {
    const size_t cnt = (n - base) < chains ? (n - base) : chains;
    for (size_t c = 0; c < cnt; ++c) {
      store(px + 4 * (base + c), X[c]);
      store(py + 4 * (base + c), Y[c]);
      store(d + 4 * (base + c), hd[c]);
    }
    if (base + chains >= n) break;
    for (size_t c = 0; c < chains; ++c) dx[c] = fsub(S.x, X[c]);
    batch_inv(dx);
    for (size_t c = 0; c < chains; ++c) {
      const U256 s = fmul(fsub(S.y, Y[c]), dx[c]);
      const U256 rx = fsub(fsub(fsqr(s), X[c]), S.x);
      Y[c] = fsub(fmul(s, fsub(X[c], rx)), Y[c]);
      X[c] = rx;
      hd[c] = mod_add(hd[c], row, ORD);
    }
  }
  return 0;
}

// ---- DP table ---------------------------------------------------------------------------------------------
struct Key128 {
  u64 h, lo, hi;
  bool operator==(const Key128 &o) const { return h == o.h && lo == o.lo && hi == o.hi; }
};
struct Key128Hash {
  size_t operator()(const Key128 &k) const { return (size_t)(k.lo * 0x9E3779B97F4A7C15ULL ^ k.hi ^ (k.h << 40)); }
};
struct Entry {
  U256 x, d;
};

// DP ingest: the reference keeps one table of sorted buckets behind one mutex and inserts the
// DPs of all GPU threads one by one (memory/hashtable.cpp:192-212, core/kangaroo.cpp:473-490).  Here the table is
// cut into NSHARD independent open-addressing tables selected by bits of x; a DP's fate depends only on earlier
// DPs with the same x, and those live in the same shard, so shards can be filled concurrently — by the worker
// threads of one large batch and by the threads of several GPUs — while the per-x order of arrival is kept.
constexpr int NSHARD = 256;
constexpr u64 CELL_USED = 1ULL << 63;  // in Entry::d.v[3]: HashTable::Convert keeps only 62 bits of that limb
// The cells of one shard: zero pages straight from the kernel (nothing to clear, and a table of several GB is not
// touched before it is used), on transparent huge pages where the kernel grants them — a DP lands on a random cell of
// a table far larger than what 4 KiB pages let the TLB cover, so with small pages every insert walks the page table.
class CellArray {
 public:
  CellArray() = default;
  explicit CellArray(size_t n) : n_(n) {
    if (!n) return;
    const size_t bytes = n * sizeof(Entry);
    if (bytes >= MAP_MIN && thp_granted()) {
      // over-allocate so that the array can start on a 2 MiB boundary, then give the slack back
      const size_t len = bytes + HUGE_SZ;
      void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m == MAP_FAILED) throw std::bad_alloc();
      const uintptr_t a = ((uintptr_t)m + HUGE_SZ - 1) & ~(uintptr_t)(HUGE_SZ - 1);
      if (a > (uintptr_t)m) munmap(m, a - (uintptr_t)m);
      const size_t tail = ((uintptr_t)m + len) - (a + bytes);
      if (tail) munmap((void *)(a + bytes), tail);
      madvise((void *)a, bytes, MADV_HUGEPAGE);
      p_ = (Entry *)a;
      mapped_ = bytes;
    } else {
      p_ = (Entry *)calloc(n, sizeof(Entry));
      if (!p_) throw std::bad_alloc();
    }
  }
  CellArray(const CellArray &) = delete;
  CellArray &operator=(const CellArray &) = delete;
  ~CellArray() { release(); }
  void swap(CellArray &o) {
    std::swap(p_, o.p_);
    std::swap(n_, o.n_);
    std::swap(mapped_, o.mapped_);
  }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  Entry &operator[](size_t i) { return p_[i]; }
  const Entry &operator[](size_t i) const { return p_[i]; }
  const Entry *begin() const { return p_; }
  const Entry *end() const { return p_ + n_; }

 private:
  static constexpr size_t HUGE_SZ = 2u << 20;
  // below one huge page a shard's array comes from the heap
  static constexpr size_t MAP_MIN = 2u << 20;
  // Fresh mmap'd memory on 4 KiB pages is the slowest way to get a table (one fault per page from 16 threads at once:
  // measured 10 x slower than calloc's recycled heap on the MI355X box), so the mapping is used only where the kernel
  // grants huge pages on request.
  static bool thp_granted() {
    static const bool yes = [] {
      char buf[128] = {0};
      FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
      if (!f) return false;
      const size_t n = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      buf[n] = 0;
      return strstr(buf, "[always]") != nullptr || strstr(buf, "[madvise]") != nullptr;
    }();
    return yes;
  }
  void release() {
    if (mapped_) munmap(p_, mapped_);
    else free(p_);
    p_ = nullptr;
    n_ = mapped_ = 0;
  }
  Entry *p_ = nullptr;
  size_t n_ = 0, mapped_ = 0;
};

struct Shard {
  static constexpr size_t FIRST_MAPPED = (2u << 20) / sizeof(Entry);
  mutable std::mutex mu;
  // One 64-byte cell per slot and nothing else (the reference's ENTRY is 64 bytes too, kangaroo.h:303-306): the key
  // is read back from the entry — x limbs 0 and 1 and the bucket number h = x.limb2 & 0x3FFFF — and bit 63 of
  // d.limb3 marks a used cell.
  CellArray cells;
  size_t used = 0;
  // A shard doubles when it is this full (in 1/256).  The shards of one table fill at the same pace; with ONE limit all
  // 256 would double during the same batch — at 84 M DPs that batch takes 250 ms instead of 50, longer than the GPU's
  // queued launch lasts.  The solver spreads the limits from 40 % to 65 %, so that the doublings are spread over the
  // batches in which the table grows by a factor 1.6.
  uint32_t max_load_256 = 154;
  static bool occupied(const Entry &c) { return (c.d.v[3] & CELL_USED) != 0; }
  static Key128 key_of(const Entry &c) { return Key128{c.x.v[2] & 0x3FFFF, c.x.v[0], c.x.v[1]}; }
  void grow() {
    // 64 KiB from the heap first (small tables stay small), then straight to one huge page, then doubling: the sizes in
    // between would each be fresh 4 KiB pages from the C library, and faulting those in costs more than the inserts
    const size_t ncap = cells.empty() ? 1024 : cells.size() < FIRST_MAPPED ? FIRST_MAPPED : cells.size() * 2;
    CellArray nc(ncap);
    // a cell's new home is its old home or that plus the old size: one scan of the old array fills the new one in two
    // ascending streams
    for (const Entry &c : cells)
      if (occupied(c)) {
        size_t j = Key128Hash()(key_of(c)) & (ncap - 1);
        while (occupied(nc[j])) j = (j + 1) & (ncap - 1);
        nc[j] = c;
      }
    cells.swap(nc);
  }
  // returns the cell of key, inserting e if absent; *existed tells which.  e.x must carry the key (key_of(e) == key).
  Entry *find_or_insert(const Key128 &key, const Entry &e, bool *existed) {
    if ((used + 1) * 256 > cells.size() * max_load_256) grow();
    size_t j = Key128Hash()(key) & (cells.size() - 1);
    while (occupied(cells[j])) {
      if (key_of(cells[j]) == key) {
        *existed = true;
        return &cells[j];
      }
      j = (j + 1) & (cells.size() - 1);
    }
    cells[j] = e;
    cells[j].d.v[3] |= CELL_USED;
    ++used;
    *existed = false;
    return &cells[j];
  }
};
// the distance word of a cell without the bookkeeping bit
static inline U256 cell_d(const Entry &c) {
  U256 d = c.d;
  d.v[3] &= ~CELL_USED;
  return d;
}

// Worker pool owned by one solver.  Round 2 created up to 3 x nthreads std::threads per batch; with one GPU thread per
// device feeding the table (core/kangaroo.cpp:780-786) that was eight producers each spawning 48 short-lived threads
// per launch.  Here the threads exist once; a batch is a sequence of parallel-for jobs on a shared queue, the
// submitting thread works on its own job as well (so eight concurrent producers are eight more workers, and a
// producer never waits for a pool that is busy with another producer's batch), and jobs are served in arrival order.
class WorkPool {
 public:
  ~WorkPool() { stop(); }
  // at least `n` pool threads (beside the callers)
  void ensure(int n) {
    std::lock_guard<std::mutex> lock(mu_);
    while ((int)th_.size() < n) th_.emplace_back([this] { loop(); });
  }
  // fn(i) for every i in [0, total); returns when all are done.  Callable from several threads at once.
  void parallel_for(int total, const std::function<void(int)> &fn) {
    if (total <= 0) return;
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->total = total;
    {
      std::lock_guard<std::mutex> lock(mu_);
      jobs_.push_back(job);
    }
    cv_work_.notify_all();
    work(*job);
    std::unique_lock<std::mutex> lock(mu_);
    cv_done_.wait(lock, [&] { return job->done.load() == job->total; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : th_) t.join();
    th_.clear();
  }

 private:
  struct Job {
    const std::function<void(int)> *fn = nullptr;
    int total = 0;
    std::atomic<int> next{0}, done{0};
  };
  void work(Job &j) {
    for (;;) {
      const int i = j.next.fetch_add(1);
      if (i >= j.total) return;
      (*j.fn)(i);
      if (j.done.fetch_add(1) + 1 == j.total) {
        std::lock_guard<std::mutex> lock(mu_);  // pairs with the wait in parallel_for
        cv_done_.notify_all();
      }
    }
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
          while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->total) jobs_.pop_front();  // fully claimed
          if (stop_ || !jobs_.empty()) break;
          cv_work_.wait(lock);
        }
        if (stop_) return;
        job = jobs_.front();
      }
      work(*job);
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::deque<std::shared_ptr<Job>> jobs_;
  std::vector<std::thread> th_;
  bool stop_ = false;
};

struct kgh_solver {
  U256 range_start, range_end, width, width2;
  int range_power;
  Aff pub, key, key_neg;
  Shard shard[NSHARD];
  std::mutex mu;  // guards found / priv / check_key
  std::atomic<bool> found{false};
  U256 priv;
  std::atomic<u64> dead{0};
  // items handed in that ended up neither in the table nor dead: the DP that solved the key, tame/wild collisions
  // that arrived after it, and whatever a batch still held when the key was found.  table + dead + unstored ==
  // items handed in, always.
  std::atomic<u64> unstored{0};
  int threads = 0;  // threads working on a large batch, the caller included (0: hardware concurrency, capped at 16)
  mutable WorkPool pool;  // also serves kgh_work_save, which takes the solver const
};
static inline int shard_of(const Key128 &k) { return (int)((k.lo * 0x9E3779B97F4A7C15ULL) >> 56); }

static void ht_convert(const U256 &x, const U256 &d, uint32_t type, Key128 &key, Entry &e) {
  // memory/hashtable.cpp:65-89
  u64 sign = 0;
  e.x = x;
  if (d.v[3] > 0x7FFFFFFFFFFFFFFFULL) {
    U256 nn = mod_neg(d, ORD);
    e.d = nn;
    e.d.v[3] &= 0x3FFFFFFFFFFFFFFFULL;
    sign = 1ULL << 63;
  } else {
    e.d = d;
    e.d.v[3] &= 0x3FFFFFFFFFFFFFFFULL;
  }
  e.d.v[1] |= sign;
  e.d.v[1] |= (u64)type << 62;
  key.h = x.v[2] & 0x3FFFF;
  key.lo = x.v[0];
  key.hi = x.v[1];
}
static void calc_dist_and_type(const U256 &D, U256 &dist, uint32_t &type) {
  // memory/hashtable.cpp:181-191
  type = (D.v[1] & 0x4000000000000000ULL) != 0;
  const bool sign = (D.v[1] & 0x8000000000000000ULL) != 0;
  dist = D;
  dist.v[1] &= 0x3FFFFFFFFFFFFFFFULL;
  if (sign) dist = mod_neg(dist, ORD);
}

static bool check_key(kgh_solver *s, U256 d1, U256 d2, int t) {
  // core/kangaroo.cpp:155-181 (the reference compares points on y only, math/point.cpp:54-56; a full
  // comparison is used here and the recovered key is verified against the public key)
  if (t & 1) d1 = mod_neg(d1, ORD);
  if (t & 2) d2 = mod_neg(d2, ORD);
  U256 pk = mod_add(d1, d2, ORD);
  Aff P = mul_g(pk);
  if (P.inf) return false;
  if (cmp(P.x, s->key.x) != 0) return false;
  if (cmp(P.y, s->key.y) != 0 && cmp(P.y, s->key_neg.y) != 0) return false;
  U256 full = mod_add(pk, reduce_once(s->range_start, ORD), ORD);
  Aff R = mul_g(full);
  if (R.inf || cmp(R.x, s->pub.x) != 0 || cmp(R.y, s->pub.y) != 0) {
    // symmetric solution: -(pk) + start
    U256 alt = mod_add(mod_neg(pk, ORD), reduce_once(s->range_start, ORD), ORD);
    Aff R2 = mul_g(alt);
    if (R2.inf || cmp(R2.x, s->pub.x) != 0 || cmp(R2.y, s->pub.y) != 0) return false;
    full = alt;
  }
  s->priv = full;
  s->found = true;
  return true;
}

// One DP into its shard.  Returns 1 = stored, 0 = duplicate / same herd (dead kangaroo), 3 = tame/wild
// collision (the caller resolves it with resolve_collision: EC work stays outside the shard lock).
// the shard's mutex is held by the caller; key / e as HashTable::Convert builds them
static int shard_insert_locked(kgh_solver *s, Shard &sh, const Key128 &key, const Entry &e, uint32_t type, U256 *other_d,
                               uint32_t *other_type) {
  bool existed;
  Entry *cell = sh.find_or_insert(key, e, &existed);
  if (!existed) return 1;
  const U256 stored = cell_d(*cell);
  if (cmp(stored, e.d) == 0) {  // ADD_DUPLICATE -> treated as a dead kangaroo
    s->dead++;
    return 0;
  }
  U256 d0;
  uint32_t t0;
  calc_dist_and_type(stored, d0, t0);
  if (t0 == type) {  // same herd
    s->dead++;
    return 0;
  }
  *other_d = d0;
  *other_type = t0;
  return 3;
}
static int shard_add(kgh_solver *s, const U256 &x, const U256 &d, u64 k_idx, U256 *other_d, uint32_t *other_type) {
  const uint32_t type = (uint32_t)(k_idx & 1);
  Key128 key;
  Entry e;
  ht_convert(x, d, type, key, e);
  Shard &sh = s->shard[shard_of(key)];
  std::lock_guard<std::mutex> lock(sh.mu);
  return shard_insert_locked(s, sh, key, e, type, other_d, other_type);
}
// CollisionCheck (core/kangaroo.cpp:183-256): returns 2 = key found, 0 = wrong collision (dead kangaroo)
static int resolve_collision(kgh_solver *s, const U256 &d, const U256 &d0, uint32_t t0) {
  std::lock_guard<std::mutex> lock(s->mu);
  if (s->found) return 2;
  const U256 &td = (t0 == 0) ? d0 : d;
  const U256 &wd = (t0 == 0) ? d : d0;
  for (int t = 0; t < 4; ++t)
    if (check_key(s, td, wd, t)) return 2;
  s->dead++;  // "Unexpected wrong collision, reset kangaroo" (core/kangaroo.cpp:206-223)
  return 0;
}
static int add_one(kgh_solver *s, const U256 &x, const U256 &d, u64 k_idx) {
  U256 d0;
  uint32_t t0;
  int r = shard_add(s, x, d, k_idx, &d0, &t0);
  if (r == 3) {
    r = resolve_collision(s, d, d0, t0);
    if (r == 2) s->unstored++;
  }
  return r;
}

kgh_solver *kgh_solver_new(const uint64_t range_start[4], const uint64_t range_end[4], const uint64_t pub_x[4],
                           const uint64_t pub_y[4]) {
  kgh_solver *s = new kgh_solver();
  for (int i = 0; i < NSHARD; ++i) s->shard[i].max_load_256 = 102 + (uint32_t)(i * 64 / (NSHARD - 1));  // 40 % .. 65 %
  s->range_start = load(range_start);
  s->range_end = load(range_end);
  sub_to(s->width, s->range_end, s->range_start);
  s->range_power = bit_length(s->width);  // InitRange, core/kangaroo.cpp:649-660
  s->width2 = shr(s->width, 1);
  s->pub = Aff{load(pub_x), load(pub_y), false};
  if (!is_zero(s->range_start)) {  // InitSearchKey, core/kangaroo.cpp:662-674
    Aff rs = mul_g(reduce_once(s->range_start, ORD));
    rs.y = mod_neg(rs.y, FP);
    s->key = aff_add(s->pub, rs);
  } else {
    s->key = s->pub;
  }
  s->key_neg = s->key;
  s->key_neg.y = mod_neg(s->key.y, FP);
  return s;
}
void kgh_solver_free(kgh_solver *s) { delete s; }
int kgh_solver_range_power(const kgh_solver *s) { return s->range_power; }
void kgh_solver_wild_offset(const kgh_solver *s, uint64_t out[4]) { store(out, s->width2); }
void kgh_solver_range_width(const kgh_solver *s, uint64_t out[4]) { store(out, s->width); }
void kgh_solver_search_key(const kgh_solver *s, uint64_t x[4], uint64_t y[4]) {
  store(x, s->key.x);
  store(y, s->key.y);
}
int kgh_solver_add(kgh_solver *s, const uint64_t x[4], const uint64_t d[4], uint64_t k_idx) {
  if (s->found) {
    s->unstored++;
    return 2;
  }
  return add_one(s, load(x), load(d), k_idx);
}
void kgh_solver_set_threads(kgh_solver *s, int threads) { s->threads = threads < 0 ? 0 : threads; }

// Batches below this size are inserted in arrival order by the calling thread (exactly the reference's
// sequence, including "stop at the DP that solves the key"); larger ones are spread over worker threads by
// shard.  In the threaded path every DP of the batch is inserted even if one of them solves the key.
constexpr uint32_t PAR_BATCH = 8192;

}  // extern "C": templates follow

// Cuts the indices 0 .. n-1 of a batch by shard, arrival order kept inside each shard: order[start[sh] .. start[sh+1])
// are the indices of shard sh, ascending.  key_lo(i) = limb 0 of item i's x.  The batch is cut into nslice consecutive
// slices, counted and scattered slice by slice on the pool.
template <class KeyLo>
static void bucket_by_shard(WorkPool &pool, int nslice, uint32_t n, KeyLo key_lo, uint32_t *order, uint8_t *sid,
                            std::vector<uint32_t> &start) {
  const uint32_t per = (n + nslice - 1) / nslice;
  std::vector<std::vector<uint32_t>> cnt(nslice, std::vector<uint32_t>(NSHARD, 0)), off(nslice, std::vector<uint32_t>(NSHARD, 0));
  pool.parallel_for(nslice, [&](int t) {
    const uint32_t lo = std::min<uint64_t>((uint64_t)t * per, n), hi = std::min<uint64_t>((uint64_t)(t + 1) * per, n);
    for (uint32_t i = lo; i < hi; ++i) {
      Key128 k;
      k.lo = key_lo(i);
      const int sh = shard_of(k);
      sid[i] = (uint8_t)sh;
      cnt[t][sh]++;
    }
  });
  start.assign(NSHARD + 1, 0);
  uint32_t run = 0;  // offsets: shard-major, slice-minor (slices are consecutive, so this is arrival order)
  for (int sh = 0; sh < NSHARD; ++sh) {
    start[sh] = run;
    for (int t = 0; t < nslice; ++t) {
      off[t][sh] = run;
      run += cnt[t][sh];
    }
  }
  start[NSHARD] = run;
  pool.parallel_for(nslice, [&](int t) {
    const uint32_t lo = std::min<uint64_t>((uint64_t)t * per, n), hi = std::min<uint64_t>((uint64_t)(t + 1) * per, n);
    for (uint32_t i = lo; i < hi; ++i) order[off[t][sid[i]]++] = i;
  });
}

// The items order[lo .. hi) of ONE shard go into it.  Every access of this loop would be a cache miss: the shard's
// items lie ~256 items apart in the caller's array, and each lands on a random cell of a table far larger than the
// caches.  So the loop is a software pipeline over runs of RUN items: the items of run r+2 are prefetched, run r+1 is
// converted and its home cells are prefetched, run r is inserted.  A chunk of CHUNK runs goes under ONE hold of the
// shard's mutex (other producers wait that long at most); arrival order inside the shard is kept.  A grow() inside a
// chunk only makes the prefetches in flight useless.
//   addr(i): address of item i (item_bytes long);  conv(i, key, entry): HashTable::Convert of item i;
//   sink(i, key, entry): inserts it, called with the shard's mutex held.
template <class Addr, class Conv, class Sink>
static void shard_pipeline(Shard &shard, const uint32_t *order, uint32_t lo, uint32_t hi, size_t item_bytes, Addr addr, Conv conv,
                           Sink sink) {
  constexpr uint32_t RUN = 16, CHUNK = 32;
  auto fetch_items = [&](uint32_t q0) {
    const uint32_t q1 = std::min<uint32_t>(q0 + RUN, hi);
    for (uint32_t q = q0; q < q1; ++q) {
      const char *p = (const char *)addr(order[q]);
      __builtin_prefetch(p, 0, 0);
      __builtin_prefetch(p + item_bytes - 1, 0, 0);
    }
  };
  Key128 keys[2][RUN];
  Entry ents[2][RUN];
  auto convert = [&](uint32_t q0, int b) {  // shard.mu held: reads the table's size
    const uint32_t q1 = std::min<uint32_t>(q0 + RUN, hi);
    for (uint32_t q = q0; q < q1; ++q) conv(order[q], keys[b][q - q0], ents[b][q - q0]);
    if (!shard.cells.empty()) {
      const size_t mask = shard.cells.size() - 1;
      for (uint32_t q = q0; q < q1; ++q) __builtin_prefetch(&shard.cells[Key128Hash()(keys[b][q - q0]) & mask], 1, 0);
    }
  };
  fetch_items(lo);
  fetch_items(lo + RUN);
  for (uint32_t c0 = lo; c0 < hi; c0 += RUN * CHUNK) {
    const uint32_t c1 = std::min<uint32_t>(c0 + RUN * CHUNK, hi);
    std::lock_guard<std::mutex> lock(shard.mu);
    int b = 0;
    convert(c0, b);
    for (uint32_t q0 = c0; q0 < c1; q0 += RUN, b ^= 1) {
      const uint32_t q1 = std::min<uint32_t>(q0 + RUN, c1);
      if (q0 + 2 * RUN < hi) fetch_items(q0 + 2 * RUN);
      if (q1 < c1) convert(q1, b ^ 1);
      for (uint32_t q = q0; q < q1; ++q) sink(order[q], keys[b][q - q0], ents[b][q - q0]);
    }
  }
}

static int solver_threads(const kgh_solver *s) {
  const int n = s->threads > 0 ? s->threads : (int)std::min<unsigned>(std::thread::hardware_concurrency(), 16u);
  return n < 1 ? 1 : n;
}

extern "C" {

int kgh_solver_add_items(kgh_solver *s, const void *items, uint32_t n, uint32_t *dead, uint32_t *n_dead) {
  const kg_item *it = (const kg_item *)items;
  uint32_t nd = 0;
  const int nthreads = solver_threads(s);
  if (n < PAR_BATCH || nthreads == 1) {
    uint32_t i = 0;
    for (; i < n && !s->found; ++i) {
      const int r = add_one(s, load(it[i].x), load(it[i].d), it[i].kIdx);
      if (r == 0 && dead) dead[nd++] = i;
    }
    s->unstored += n - i;  // what the batch still held when the key was found
    if (n_dead) *n_dead = nd;
    return s->found ? 2 : 0;
  }
  s->pool.ensure(nthreads - 1);  // the calling thread is the nthreads-th worker
  // 1. bucket the item indices by shard.  Staging that outlives the call (one set per calling thread, i.e. per GPU): at
  //    8.4 M items a launch these are 42 MB, and fresh pages for them on every call cost more than filling them
  static thread_local std::vector<uint32_t> order_tl;
  static thread_local std::vector<uint8_t> sid_tl;
  if (order_tl.size() < n) order_tl.resize(n);
  if (sid_tl.size() < n) sid_tl.resize(n);
  uint32_t *const order = order_tl.data();  // the workers must see THIS thread's arrays: a lambda that named the
  uint8_t *const sid = sid_tl.data();       // thread_local vectors would get the executing thread's own
  std::vector<uint32_t> start;
  bucket_by_shard(s->pool, nthreads, n, [&](uint32_t i) { return it[i].x[0]; }, order, sid, start);
  // 2. shards in parallel; dead kangaroos and tame/wild collisions are recorded per shard
  struct Coll {
    uint32_t idx;
    U256 d0;
    uint32_t t0;
  };
  std::vector<std::vector<uint32_t>> sdead(NSHARD);
  std::vector<std::vector<Coll>> scoll(NSHARD);
  s->pool.parallel_for(NSHARD, [&](int sh) {
    Shard &shard = s->shard[sh];
    shard_pipeline(
        shard, order, start[sh], start[sh + 1], sizeof(kg_item), [&](uint32_t i) { return (const void *)&it[i]; },
        [&](uint32_t i, Key128 &key, Entry &e) { ht_convert(load(it[i].x), load(it[i].d), (uint32_t)(it[i].kIdx & 1), key, e); },
        [&](uint32_t i, const Key128 &key, const Entry &e) {
          U256 d0;
          uint32_t t0;
          const int r = shard_insert_locked(s, shard, key, e, (uint32_t)(it[i].kIdx & 1), &d0, &t0);
          if (r == 0) sdead[sh].push_back(i);
          else if (r == 3) scoll[sh].push_back(Coll{i, d0, t0});
        });
  });
  // 3. collisions between herds, in arrival order (EC arithmetic, rare)
  std::vector<Coll> coll;
  for (auto &v : scoll) coll.insert(coll.end(), v.begin(), v.end());
  std::sort(coll.begin(), coll.end(), [](const Coll &a, const Coll &b) { return a.idx < b.idx; });
  std::vector<uint32_t> all_dead;
  for (auto &v : sdead) all_dead.insert(all_dead.end(), v.begin(), v.end());
  for (const Coll &c : coll) {
    if (resolve_collision(s, load(it[c.idx].d), c.d0, c.t0) == 0) all_dead.push_back(c.idx);
    else s->unstored++;  // the DP that solves the key, or a collision that arrived after it
  }
  std::sort(all_dead.begin(), all_dead.end());
  if (dead)
    for (uint32_t i : all_dead) dead[nd++] = i;
  if (n_dead) *n_dead = nd;
  return s->found ? 2 : 0;
}
int kgh_solver_found(const kgh_solver *s, uint64_t priv[4]) {
  if (!s->found) return 0;
  if (priv) store(priv, s->priv);
  return 1;
}
uint64_t kgh_solver_table_size(const kgh_solver *s) {
  uint64_t n = 0;
  for (const Shard &sh : s->shard) {
    std::lock_guard<std::mutex> lock(sh.mu);
    n += sh.used;
  }
  return n;
}
uint64_t kgh_solver_dead_count(const kgh_solver *s) { return s->dead; }
uint64_t kgh_solver_unstored_count(const kgh_solver *s) { return s->unstored; }
uint64_t kgh_solver_dump(const kgh_solver *s, uint64_t *out, uint64_t cap) {
  uint64_t n = 0;
  for (const Shard &sh : s->shard) {
    std::lock_guard<std::mutex> lock(sh.mu);  // a concurrent insert may grow (reallocate) the shard
    for (const Entry &c : sh.cells) {
      if (!Shard::occupied(c)) continue;
      if (n >= cap) return n;
      const U256 d = cell_d(c);
      out[9 * n] = Shard::key_of(c).h;
      memcpy(out + 9 * n + 1, c.x.v, 32);
      memcpy(out + 9 * n + 5, d.v, 32);
      ++n;
    }
  }
  return n;
}


// ---- work file ---------------------------------------------------------------------------------
// Layout of the reference's "full work file" (HEADW), all little endian:
//   u32 0xFA6A8001 (constants.h:43), u32 version 0                       storage/backup.cpp:256-271
//   u32 dpSize, rangeStart[32], rangeEnd[32], key.x[32], key.y[32], u64 totalCount, f64 totalTime
//                                                                         kangaroo.h:263-287, backup.cpp:273-286
//   for each of the 2^18 buckets h: u32 nbItem, u32 maxItem, nbItem x { x[16], d[16] }
//       (memory/hashtable.cpp:13-21,262-285), items ascending by (x.i64[1], x.i64[0]) (hashtable.cpp:192-226)
//   u64 totalWalk, then totalWalk x { px[32], py[32], d[32] }           storage/backup.cpp:389-410
// maxItem is the capacity of the reference's std::vector; a table filled by single inserts has the next power of
// two, which is what is written here (any value >= nbItem loads).
// PINNED (round 3): tests/golden/ref_work_r32_dp8.bin.gz was written by the reference's own Kangaroo::SaveWork
// (oracle/ref_harness.cpp --savework); kgh_work_load reads it, kgh_work_save reproduces it byte for byte, and the
// reference's LoadWork + FetchWalks read a file written here (tests/test_workfile.py).
namespace {
constexpr uint32_t HEADW_MAGIC = 0xFA6A8001u;
constexpr uint32_t HASH_BUCKETS = 1u << 18;
struct FileEntry {
  u64 x[2], d[2];
};
bool wr(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }
}  // namespace

int kgh_work_save(const kgh_solver *s, const char *path, uint32_t dp_size, uint64_t total_count, double total_time,
                  const uint64_t *px, const uint64_t *py, const uint64_t *d, uint64_t n_kangaroo) {
  if (!s || !path) return -1;
  FILE *f = fopen(path, "wb");
  if (!f) return -2;
  const uint32_t version = 0;
  bool ok = wr(f, &HEADW_MAGIC, 4) && wr(f, &version, 4) && wr(f, &dp_size, 4) && wr(f, s->range_start.v, 32) &&
            wr(f, s->range_end.v, 32) && wr(f, s->pub.x.v, 32) && wr(f, s->pub.y.v, 32) && wr(f, &total_count, 8) &&
            wr(f, &total_time, 8);
  // Gather the shards into the reference's 2^18 buckets, on the solver's worker pool (round 3: a save of 49 M entries
  // took 37 s on one thread, with every GPU thread parked for as long).  Each shard is copied out under its mutex in
  // ONE scan (the reference saves with every worker parked, storage/backup.cpp:327-345; this must also be safe if a
  // producer is still inserting: a concurrent grow() reallocates the cells).  The copies are cut into 256 partitions
  // of 1024 consecutive buckets; every partition is sorted by (bucket, x) and laid out as it goes into the file.
  struct Snap {
    uint32_t h;
    FileEntry e;
  };
  constexpr int NPART = 256, PART_SHIFT = 10;  // HASH_BUCKETS = NPART << PART_SHIFT
  const int nthreads = solver_threads(s);
  s->pool.ensure(nthreads - 1);
  std::vector<std::vector<Snap>> per(NSHARD);
  std::vector<std::vector<uint64_t>> off(NSHARD, std::vector<uint64_t>(NPART, 0));
  s->pool.parallel_for(NSHARD, [&](int i) {
    const Shard &sh = s->shard[i];
    std::lock_guard<std::mutex> lock(sh.mu);
    per[i].reserve(sh.used);
    for (const Entry &c : sh.cells)
      if (Shard::occupied(c)) {
        const uint32_t h = (uint32_t)Shard::key_of(c).h;
        per[i].push_back(Snap{h, FileEntry{{c.x.v[0], c.x.v[1]}, {c.d.v[0], c.d.v[1]}}});
        off[i][h >> PART_SHIFT]++;
      }
  });
  std::vector<uint64_t> pstart(NPART + 1, 0);
  {
    uint64_t run = 0;  // partition-major, shard-minor
    for (int p = 0; p < NPART; ++p) {
      pstart[p] = run;
      for (int i = 0; i < NSHARD; ++i) {
        const uint64_t c = off[i][p];
        off[i][p] = run;
        run += c;
      }
    }
    pstart[NPART] = run;
  }
  std::vector<Snap> parted(pstart[NPART]);
  s->pool.parallel_for(NSHARD, [&](int i) {
    for (const Snap &q : per[i]) parted[off[i][q.h >> PART_SHIFT]++] = q;
    std::vector<Snap>().swap(per[i]);
  });
  std::vector<std::vector<char>> image(NPART);  // the partitions as they lie in the file
  s->pool.parallel_for(NPART, [&](int p) {
    Snap *b = parted.data() + pstart[p], *e = parted.data() + pstart[p + 1];
    std::sort(b, e, [](const Snap &l, const Snap &r) {
      return l.h != r.h ? l.h < r.h : l.e.x[1] != r.e.x[1] ? l.e.x[1] < r.e.x[1] : l.e.x[0] < r.e.x[0];
    });
    std::vector<char> &out = image[p];
    out.resize(((size_t)8 << PART_SHIFT) + (size_t)(e - b) * sizeof(FileEntry));
    char *w = out.data();
    for (uint32_t h = (uint32_t)p << PART_SHIFT; h < ((uint32_t)p + 1) << PART_SHIFT; ++h) {
      const Snap *q = b;
      while (q < e && q->h == h) ++q;
      const uint32_t nb = (uint32_t)(q - b);
      uint32_t cap = 0;  // what the reference's vector holds after nb single inserts
      if (nb) {
        cap = 1;
        while (cap < nb) cap <<= 1;
      }
      memcpy(w, &nb, 4);
      memcpy(w + 4, &cap, 4);
      w += 8;
      for (; b < q; ++b, w += sizeof(FileEntry)) memcpy(w, &b->e, sizeof(FileEntry));
    }
  });
  std::vector<Snap>().swap(parted);
  for (int p = 0; p < NPART && ok; ++p) {
    ok = wr(f, image[p].data(), image[p].size());
    std::vector<char>().swap(image[p]);
  }
  const uint64_t walks = (px && py && d) ? n_kangaroo : 0;
  ok = ok && wr(f, &walks, 8);
  for (uint64_t i = 0; i < walks && ok; ++i) ok = wr(f, px + 4 * i, 32) && wr(f, py + 4 * i, 32) && wr(f, d + 4 * i, 32);
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : -3;
}

int kgh_work_info(const char *path, uint32_t *dp_size, uint64_t range_start[4], uint64_t range_end[4], uint64_t key_x[4],
                  uint64_t key_y[4], uint64_t *total_count, double *total_time, uint64_t *n_items, uint64_t *n_kangaroo) {
  FILE *f = fopen(path, "rb");
  if (!f) return -2;
  uint32_t magic = 0, version = 0, dp = 0;
  u64 rs[4], re[4], kx[4], ky[4], tc = 0, items = 0, walks = 0;
  double tt = 0;
  bool ok = rd(f, &magic, 4) && rd(f, &version, 4) && magic == HEADW_MAGIC && rd(f, &dp, 4) && rd(f, rs, 32) && rd(f, re, 32) &&
            rd(f, kx, 32) && rd(f, ky, 32) && rd(f, &tc, 8) && rd(f, &tt, 8);
  for (uint32_t h = 0; h < HASH_BUCKETS && ok; ++h) {  // HashTable::SeekNbItem, memory/hashtable.cpp:296-306
    uint32_t nb, cap;
    ok = rd(f, &nb, 4) && rd(f, &cap, 4) && fseeko(f, (off_t)nb * 32, SEEK_CUR) == 0;
    items += nb;
  }
  ok = ok && rd(f, &walks, 8);
  fclose(f);
  if (!ok) return -4;
  if (dp_size) *dp_size = dp;
  if (range_start) memcpy(range_start, rs, 32);
  if (range_end) memcpy(range_end, re, 32);
  if (key_x) memcpy(key_x, kx, 32);
  if (key_y) memcpy(key_y, ky, 32);
  if (total_count) *total_count = tc;
  if (total_time) *total_time = tt;
  if (n_items) *n_items = items;
  if (n_kangaroo) *n_kangaroo = walks;
  return 0;
}

// Loads the table of a work file into s (which must have been created for the same range and key: checked) and
// up to cap kangaroos into px/py/d.  Kangaroo::LoadWork + FetchWalks (storage/backup.cpp:54-253).
int kgh_work_load(kgh_solver *s, const char *path, uint64_t *px, uint64_t *py, uint64_t *d, uint64_t cap,
                  uint64_t *n_loaded) {
  if (!s || !path) return -1;
  FILE *f = fopen(path, "rb");
  if (!f) return -2;
  uint32_t magic = 0, version = 0, dp = 0;
  u64 rs[4], re[4], kx[4], ky[4], tc = 0, walks = 0;
  double tt = 0;
  bool ok = rd(f, &magic, 4) && rd(f, &version, 4) && magic == HEADW_MAGIC && rd(f, &dp, 4) && rd(f, rs, 32) && rd(f, re, 32) &&
            rd(f, kx, 32) && rd(f, ky, 32) && rd(f, &tc, 8) && rd(f, &tt, 8);
  if (ok && (memcmp(rs, s->range_start.v, 32) || memcmp(re, s->range_end.v, 32) || memcmp(kx, s->pub.x.v, 32) ||
             memcmp(ky, s->pub.y.v, 32))) {
    fclose(f);
    return -5;  // another range or key
  }
  // The buckets are read in order; every LOAD_CHUNK entries go into the table on the solver's worker pool, cut by shard
  // like a batch of DPs (HashTable::LoadTable reads entry by entry, memory/hashtable.cpp:306-320).
  struct Rec {
    uint32_t h;
    FileEntry e;
  };
  constexpr size_t LOAD_CHUNK = 8u << 20;
  const int nthreads = solver_threads(s);
  s->pool.ensure(nthreads - 1);
  std::vector<Rec> recs;
  std::vector<FileEntry> buf;
  std::vector<uint32_t> order, start;
  std::vector<uint8_t> sid;
  auto flush = [&]() {
    const uint32_t n = (uint32_t)recs.size();
    if (!n) return;
    order.resize(n);
    sid.resize(n);
    bucket_by_shard(s->pool, nthreads, n, [&](uint32_t i) { return recs[i].e.x[0]; }, order.data(), sid.data(), start);
    s->pool.parallel_for(NSHARD, [&](int sh) {
      Shard &shard = s->shard[sh];
      shard_pipeline(
          shard, order.data(), start[sh], start[sh + 1], sizeof(Rec), [&](uint32_t i) { return (const void *)&recs[i]; },
          [&](uint32_t i, Key128 &key, Entry &e) {
            const Rec &r = recs[i];
            key = Key128{(u64)r.h, r.e.x[0], r.e.x[1]};
            e.x = U256{{r.e.x[0], r.e.x[1], (u64)r.h, 0}};  // the file keeps 128 bits of x; limb 2 carries the bucket number
            e.d = U256{{r.e.d[0], r.e.d[1], 0, 0}};
          },
          [&](uint32_t, const Key128 &key, const Entry &e) {
            bool existed;
            shard.find_or_insert(key, e, &existed);
          });
    });
    recs.clear();
  };
  for (uint32_t h = 0; h < HASH_BUCKETS && ok; ++h) {
    uint32_t nb, capb;
    ok = rd(f, &nb, 4) && rd(f, &capb, 4);
    if (!ok || !nb) continue;
    buf.resize(nb);
    ok = rd(f, buf.data(), (size_t)nb * sizeof(FileEntry));
    if (!ok) break;
    for (uint32_t i = 0; i < nb; ++i) recs.push_back(Rec{h, buf[i]});
    if (recs.size() >= LOAD_CHUNK) flush();
  }
  if (ok) flush();
  ok = ok && rd(f, &walks, 8);
  uint64_t got = 0;
  for (; ok && got < walks && got < cap && px && py && d; ++got)
    ok = rd(f, px + 4 * got, 32) && rd(f, py + 4 * got, 32) && rd(f, d + 4 * got, 32);
  fclose(f);
  if (n_loaded) *n_loaded = got;
  return ok ? 0 : -4;
}

}  // extern "C"
