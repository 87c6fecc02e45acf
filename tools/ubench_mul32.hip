// Micro-benchmark: a field multiplication mod p on 8 x 32-bit saturated limbs — v_mad_u64_u32 with its carry-out
// caught by v_addc_co_u32 into a third accumulator word, the form BASELINE's north star sketches ("4x64 / 8x32
// limbs via v_mad_u64_u32 chains") — against the 9 x 29-bit multiplier the kernels use (fe29.h), as MEASURED
// bodies: chains of dependent multiplications at 2 waves per SIMD on every CU, sustained for about a second, with
// the shader clock and board power of each.  Both are checked against a plain big-integer product on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ab/ubench_mul32 tools/ubench_mul32.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../kangaroo_amd/csrc/fe29.h"
using namespace kg;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct fe32 {
  u32 w[8];
};

// acc (64 bit) += a * b, carry-out added into acc2: two instructions per partial product
__device__ __forceinline__ void mac96(u64 &acc, u32 &acc2, u32 a, u32 b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
}
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// r = a * b mod p, lazily reduced like the reference (value < 2^256, canonical unless it lands in [p, 2^256))
__device__ __forceinline__ void fe32_mul(fe32 &r, const fe32 &a, const fe32 &b) {
  u32 t[16];
  u64 acc = 0;
  u32 acc2 = 0;
  sfor<0, 15>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int lo = k > 7 ? k - 7 : 0, hi = k < 7 ? k : 7;
    sfor<lo, hi + 1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mac96(acc, acc2, a.w[i], b.w[k - i]);
    });
    t[k] = (u32)acc;
    acc = (acc >> 32) | ((u64)acc2 << 32);
    acc2 = 0;
  });
  t[15] = (u32)acc;
  // fold the high half: 2^256 = 2^32 + 977 (mod p).  low_j + 977 * h_j + h_(j-1), carries through a 64-bit chain
  u64 c = 0;
  u32 s[9];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    c += (u64)t[j] + (j > 0 ? t[8 + j - 1] : 0u);
    c = mad_u64_k(t[8 + j], 977u, c);
    s[j] = (u32)c;
    c >>= 32;
  }
  c += t[15];
  s[8] = (u32)c;  // < 2^33 in all: bits 256 and up, folded once more
  const u64 top = c;
  u64 d = (u64)s[0] + top * 977u;
  r.w[0] = (u32)d;
  d = (d >> 32) + s[1] + top;
  r.w[1] = (u32)d;
  d >>= 32;
#pragma unroll
  for (int j = 2; j < 8; ++j) {
    d += s[j];
    r.w[j] = (u32)d;
    d >>= 32;
  }
  // a carry out of bit 256 here has probability about 2^-190 (the reference drops it too, intmod.cpp:591-592)
}

// ---- third form (round 3): double-precision FMA products, 5 limbs of 52 bits ------------------------------------
// Only the PRODUCT PHASE of a multiplication: the 25 limb products split exactly into high and low halves
//   hi = fma_rz(a, b, 2^104)                    = 2^104 + floor(a b / 2^52) 2^52     (exponent fixed: [2^104, 2^105))
//   lo = fma   (a, b, (2^104 + 2^52) - hi)      = 2^52 + (a b mod 2^52)              (exponent fixed: [2^52, 2^53))
// and accumulated as 64-bit INTEGERS through their bit patterns into ten column sums (Emmart, Zheng, Weems, "Faster
// modular exponentiation using double precision floating point arithmetic on the GPU", ARITH 2018).  No carries, no
// reduction mod p, no conversion to or from packed words: a lower bound on what a 5 x 52 field multiplication costs.
struct fe52 {
  double l[5];
};
__device__ __forceinline__ void mul52_columns(u64 (&col)[10], const fe52 &a, const fe52 &b) {
  const double C1 = 0x1p104, C3 = 0x1p104 + 0x1p52;
  constexpr u64 B104 = 0x4670000000000000ull, B52 = 0x4330000000000000ull;  // bit patterns of 2^104 and 2^52
  // column k receives the low halves of the products with i + j = k and the high halves of those with i + j = k - 1;
  // the accumulators start at minus the biases those bit patterns carry
  constexpr int NLO[10] = {1, 2, 3, 4, 5, 4, 3, 2, 1, 0};
#pragma unroll
  for (int k = 0; k < 10; ++k) col[k] = 0ull - (u64)NLO[k] * B52 - (u64)(k > 0 ? NLO[k - 1] : 0) * B104;
  sfor<0, 5>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    sfor<0, 5>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double h = __builtin_fma(a.l[i], b.l[j], C1);       // round-toward-zero mode (set by the kernel)
      const double l = __builtin_fma(a.l[i], b.l[j], C3 - h);   // exact
      col[i + j + 1] += (u64)__double_as_longlong(h);
      col[i + j] += (u64)__double_as_longlong(l);
    });
  });
}
__device__ __forceinline__ double limb52(u64 v) {  // integer < 2^52 -> the same value as a double (exact)
  return __longlong_as_double((long long)((v & ((1ull << 52) - 1)) | 0x4330000000000000ull)) - 0x1p52;
}

template <int KIND>
__global__ void __launch_bounds__(256, 2) k_chain(u32 *out, const u32 *in, u32 iters, u64 *clk) {
  const u32 T = blockIdx.x * 256 + threadIdx.x;
  const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (KIND == 2) {  // 5 x 52 in doubles, product phase only
    // MODE.fp_round[3:2] (double precision) = toward zero.  As inline asm: the compiler's mode-register pass tracks
    // s_setreg instructions it can see and "restores" round-to-nearest in front of the next double-precision operation
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    fe52 x, y;
    for (int i = 0; i < 5; ++i) {
      const u64 vx = ((u64)(in[i] ^ (T * 2654435761u * (i + 1))) << 20) ^ in[8 + i];
      const u64 vy = ((u64)(in[8 + i] + T) << 21) ^ in[i];
      x.l[i] = limb52(vx);
      y.l[i] = limb52(vy);
    }
    u64 col[10];
    for (u32 it = 0; it < iters; ++it) {
      mul52_columns(col, x, y);
      if (iters == 1) break;  // correctness launch: the raw column sums are written out below
#pragma unroll
      for (int i = 0; i < 5; ++i) x.l[i] = limb52(col[i] ^ col[i + 5]);   // a dependent, well mixed next operand
      mul52_columns(col, y, x);
#pragma unroll
      for (int i = 0; i < 5; ++i) y.l[i] = limb52(col[i] ^ (col[i + 5] >> 3));
    }
    if (iters == 1) {
      for (int k = 0; k < 10; ++k) {  // 16 words per lane are reserved: columns 0..7 here, 8 and 9 folded into a check word each
        if (k < 8) {
          out[(size_t)T * 16 + 2 * k] = (u32)col[k];
          out[(size_t)T * 16 + 2 * k + 1] = (u32)(col[k] >> 32);
        }
      }
    } else {
      u64 acc = 0;
      for (int i = 0; i < 5; ++i) acc ^= (u64)__double_as_longlong(x.l[i]) ^ (u64)__double_as_longlong(y.l[i]);
      out[(size_t)T * 16] = (u32)acc;
      out[(size_t)T * 16 + 1] = (u32)(acc >> 32);
    }
    // (the rounding mode is not restored: the compiler is free to move such a restore in front of the FMAs, which do
    // not depend on it in its model, and the wave ends here anyway)
  } else if constexpr (KIND == 0) {  // 9 x 29
    fe x, y;
    u32 wa[8], wb[8];
    for (int i = 0; i < 8; ++i) {
      wa[i] = in[i] ^ (T * 2654435761u * (i + 1));
      wb[i] = in[8 + i] + T;
    }
    wa[7] >>= 1;
    wb[7] >>= 1;
    fe_from_words(x, wa);
    fe_from_words(y, wb);
    for (u32 it = 0; it < iters; ++it) {
      fe_mul(x, x, y);
      fe_mul(y, y, x);
    }
    fe_canon(x);
    fe_canon(y);
    fe_to_words(wa, x);
    fe_to_words(wb, y);
    for (int i = 0; i < 8; ++i) {
      out[(size_t)T * 16 + i] = wa[i];
      out[(size_t)T * 16 + 8 + i] = wb[i];
    }
  } else {  // 8 x 32
    fe32 x, y;
    for (int i = 0; i < 8; ++i) {
      x.w[i] = in[i] ^ (T * 2654435761u * (i + 1));
      y.w[i] = in[8 + i] + T;
    }
    x.w[7] >>= 1;
    y.w[7] >>= 1;
    for (u32 it = 0; it < iters; ++it) {
      fe32_mul(x, x, y);
      fe32_mul(y, y, x);
    }
    for (int i = 0; i < 8; ++i) {
      out[(size_t)T * 16 + i] = x.w[i];
      out[(size_t)T * 16 + 8 + i] = y.w[i];
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (T == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

// ---- host big integers (little-endian 32-bit words) for the check ----
typedef std::vector<u32> big;
static big mul(const big &a, const big &b) {
  big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); ++i) {
    u64 c = 0;
    for (size_t j = 0; j < b.size(); ++j) {
      c += (u64)a[i] * b[j] + r[i + j];
      r[i + j] = (u32)c;
      c >>= 32;
    }
    r[i + b.size()] = (u32)c;
  }
  return r;
}
static big mod_p(big v) {  // v < 2^512 -> canonical mod p = 2^256 - 2^32 - 977
  for (int round = 0; round < 3; ++round) {
    big lo(v.begin(), v.begin() + 8), hi(v.begin() + 8, v.end());
    big k = {977u, 1u};
    big f = mul(hi, k);
    f.resize(std::max<size_t>(f.size(), 9) + 1, 0);
    u64 c = 0;
    for (size_t i = 0; i < f.size(); ++i) {
      c += (u64)f[i] + (i < 8 ? lo[i] : 0u);
      f[i] = (u32)c;
      c >>= 32;
    }
    v = f;
    v.resize(16, 0);
  }
  const u32 P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  big r(v.begin(), v.begin() + 8);
  for (int t = 0; t < 2; ++t) {
    bool ge = true;
    for (int i = 7; i >= 0; --i)
      if (r[i] != P[i]) {
        ge = r[i] > P[i];
        break;
      }
    if (!ge) break;
    long long br = 0;
    for (int i = 0; i < 8; ++i) {
      long long d = (long long)r[i] - P[i] - br;
      br = d < 0;
      r[i] = (u32)(d + (br ? (1LL << 32) : 0));
    }
  }
  return r;
}

static std::string find_power_file() {
  for (int card = 0; card < 16; ++card)
    for (int hw = 0; hw < 16; ++hw)
      for (const char *leaf : {"power1_average", "power1_input"}) {
        char p[256];
        snprintf(p, sizeof p, "/sys/class/drm/card%d/device/hwmon/hwmon%d/%s", card, hw, leaf);
        FILE *f = fopen(p, "r");
        if (f) {
          fclose(f);
          return p;
        }
      }
  return "";
}

template <int KIND>
static void run(const char *name, double target_s, const std::string &pfile) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 2;
  u32 *out, *in;
  u64 *clk;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 16 * 4));
  CHECK(hipMalloc(&in, 64));
  CHECK(hipMalloc(&clk, 16));
  u32 hin[16];
  for (int i = 0; i < 16; ++i) hin[i] = 0x9E3779B9u * (i + 1) + 12345u;
  CHECK(hipMemcpy(in, hin, 64, hipMemcpyHostToDevice));
  // correctness: 3 iterations (one for the FP64 product phase), lanes 0..255 against the host
  hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, KIND == 2 ? 1u : 3u, clk);
  CHECK(hipDeviceSynchronize());
  std::vector<u32> h(256 * 16);
  CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (u32 T = 0; T < 256 && KIND == 2; ++T) {
    // the low eight column sums of the 5 x 5 limb product, exactly: col[k] = sum lo(a_i b_j, i+j=k) + sum hi(a_i b_j, i+j=k-1)
    u64 a[5], b[5], want[10] = {0};
    for (int i = 0; i < 5; ++i) {
      a[i] = ((((u64)(hin[i] ^ (T * 2654435761u * (i + 1))) << 20) ^ hin[8 + i])) & ((1ull << 52) - 1);
      b[i] = ((((u64)(hin[8 + i] + T) << 21) ^ hin[i])) & ((1ull << 52) - 1);
    }
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) {
        const unsigned __int128 pr = (unsigned __int128)a[i] * b[j];
        want[i + j] += (u64)(pr & (((unsigned __int128)1 << 52) - 1));
        want[i + j + 1] += (u64)(pr >> 52);
      }
    for (int k = 0; k < 8; ++k) {
      const u64 got = (u64)h[T * 16 + 2 * k] | ((u64)h[T * 16 + 2 * k + 1] << 32);
      bad += got != want[k];
    }
  }
  for (u32 T = 0; T < 256 && KIND != 2; ++T) {
    big x(8), y(8);
    for (int i = 0; i < 8; ++i) {
      x[i] = hin[i] ^ (T * 2654435761u * (i + 1));
      y[i] = hin[8 + i] + T;
    }
    x[7] >>= 1;
    y[7] >>= 1;
    for (int it = 0; it < 3; ++it) {
      x = mod_p(mul(x, y));
      y = mod_p(mul(y, x));
    }
    big gx(h.begin() + T * 16, h.begin() + T * 16 + 8), gy(h.begin() + T * 16 + 8, h.begin() + T * 16 + 16);
    gx.resize(16, 0);
    gy.resize(16, 0);
    gx = mod_p(gx);  // the 8 x 32 form returns lazily reduced values
    gy = mod_p(gy);
    bad += (gx != x) || (gy != y);
  }
  // timing
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  u32 iters = 2000;
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  iters = (u32)(iters * target_s * 1e3 / ms);
  std::atomic<bool> stop{false};
  std::vector<double> w;
  std::thread th([&] {
    while (!stop) {
      FILE *f = pfile.empty() ? nullptr : fopen(pfile.c_str(), "r");
      long long uw = 0;
      if (f) {
        if (fscanf(f, "%lld", &uw) != 1) uw = 0;
        fclose(f);
      }
      w.push_back(uw * 1e-6);
      std::this_thread::sleep_for(std::chrono::milliseconds(25));
    }
  });
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  stop = true;
  th.join();
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  u64 hc[2];
  CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
  std::vector<double> half(w.begin() + w.size() / 2, w.end());
  std::sort(half.begin(), half.end());
  const double watts = half.empty() ? 0 : half[half.size() / 2];
  const double muls = 2.0 * iters * blocks * 256;
  printf("%-34s check %s   %7.2f G mul/s   %6.0f MHz   %5.0f W   %6.1f cycles per mul per wave at 2 waves/SIMD   (%.2f s)\n", name,
         bad ? "FAILED" : "ok", muls / (ms * 1e-3) / 1e9, hc[1] ? 100.0 * hc[0] / hc[1] : 0.0, watts, (double)hc[0] / (2.0 * iters) / 2.0, ms * 1e-3);
  if (bad) exit(1);
}

int main(int argc, char **argv) {
  const double target_s = argc > 1 ? atof(argv[1]) : 1.0;
  const std::string pfile = find_power_file();
  run<0>("fe_mul, 9 x 29-bit limbs (fe29.h)", target_s, pfile);
  run<1>("8 x 32-bit limbs, mad + addc carry", target_s, pfile);
  run<2>("5 x 52 FP64, PRODUCT PHASE ONLY", target_s, pfile);
  return 0;
}
