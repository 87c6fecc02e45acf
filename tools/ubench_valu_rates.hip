// Integer VALU micro-benchmarks for gfx950: issue rates of v_mad_u64_u32 and friends.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32;
typedef uint64_t u64;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define ITERS 2048

// 8 independent mad chains, each dependent on itself
__global__ void k_mad_indep8(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u64_u32 %0, s[4:5], %8, %9, %0\n\t"
        "v_mad_u64_u32 %1, s[4:5], %8, %9, %1\n\t"
        "v_mad_u64_u32 %2, s[4:5], %8, %9, %2\n\t"
        "v_mad_u64_u32 %3, s[4:5], %8, %9, %3\n\t"
        "v_mad_u64_u32 %4, s[4:5], %8, %9, %4\n\t"
        "v_mad_u64_u32 %5, s[4:5], %8, %9, %5\n\t"
        "v_mad_u64_u32 %6, s[4:5], %8, %9, %6\n\t"
        "v_mad_u64_u32 %7, s[4:5], %8, %9, %7\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a), "v"(b) : "s4", "s5");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// one dependent chain (8 per iteration to match counts)
__global__ void k_mad_dep(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 c0 = a;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, s[4:5], %1, %2, %0\n\t"
        : "+v"(c0) : "v"(a), "v"(b) : "s4", "s5");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
}

// v_mul_lo_u32 independent
__global__ void k_mullo(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mul_lo_u32 %0, %0, %8\n\t"
        "v_mul_lo_u32 %1, %1, %8\n\t"
        "v_mul_lo_u32 %2, %2, %8\n\t"
        "v_mul_lo_u32 %3, %3, %8\n\t"
        "v_mul_lo_u32 %4, %4, %8\n\t"
        "v_mul_lo_u32 %5, %5, %8\n\t"
        "v_mul_lo_u32 %6, %6, %8\n\t"
        "v_mul_lo_u32 %7, %7, %8\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// v_mul_hi_u32 independent
__global__ void k_mulhi(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mul_hi_u32 %0, %0, %8\n\t"
        "v_mul_hi_u32 %1, %1, %8\n\t"
        "v_mul_hi_u32 %2, %2, %8\n\t"
        "v_mul_hi_u32 %3, %3, %8\n\t"
        "v_mul_hi_u32 %4, %4, %8\n\t"
        "v_mul_hi_u32 %5, %5, %8\n\t"
        "v_mul_hi_u32 %6, %6, %8\n\t"
        "v_mul_hi_u32 %7, %7, %8\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// v_mad_u32_u24 independent (expected full rate)
__global__ void k_mad24(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u32_u24 %0, %0, %8, %0\n\t"
        "v_mad_u32_u24 %1, %1, %8, %1\n\t"
        "v_mad_u32_u24 %2, %2, %8, %2\n\t"
        "v_mad_u32_u24 %3, %3, %8, %3\n\t"
        "v_mad_u32_u24 %4, %4, %8, %4\n\t"
        "v_mad_u32_u24 %5, %5, %8, %5\n\t"
        "v_mad_u32_u24 %6, %6, %8, %6\n\t"
        "v_mad_u32_u24 %7, %7, %8, %7\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// v_add_u32 independent (full rate reference)
__global__ void k_add(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_add_u32 %0, %0, %8\n\t"
        "v_add_u32 %1, %1, %8\n\t"
        "v_add_u32 %2, %2, %8\n\t"
        "v_add_u32 %3, %3, %8\n\t"
        "v_add_u32 %4, %4, %8\n\t"
        "v_add_u32 %5, %5, %8\n\t"
        "v_add_u32 %6, %6, %8\n\t"
        "v_add_u32 %7, %7, %8\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// v_lshl_add_u64 independent
__global__ void k_add64(u64 *out, u32 seed) {
  u64 a = threadIdx.x * 2654435761u + seed;
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_lshl_add_u64 %0, %0, 0, %8\n\t"
        "v_lshl_add_u64 %1, %1, 0, %8\n\t"
        "v_lshl_add_u64 %2, %2, 0, %8\n\t"
        "v_lshl_add_u64 %3, %3, 0, %8\n\t"
        "v_lshl_add_u64 %4, %4, 0, %8\n\t"
        "v_lshl_add_u64 %5, %5, 0, %8\n\t"
        "v_lshl_add_u64 %6, %6, 0, %8\n\t"
        "v_lshl_add_u64 %7, %7, 0, %8\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// carry chain: v_add_co + 7 v_addc with required s_nop 1 between (hazard: VALU writes VCC -> VALU reads VCC, 2 wait states)
__global__ void k_carry_nop(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_add_co_u32 %0, vcc, %0, %8\n\ts_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %7, vcc, %7, %8, vcc\n\ts_nop 1\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// carry chain without nops: measures whether HW interlocks (timing only; results may be wrong if no interlock)
__global__ void k_carry_nonop(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %8, vcc\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// mad + addc pairs with rotating carry SGPRs (2 wait states respected by software pipelining)
__global__ void k_mad_addc_rot(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 acc = a; u32 h = 0;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u64_u32 %0, s[4:5], %2, %3, %0\n\t"
        "v_mad_u64_u32 %0, s[6:7], %2, %3, %0\n\t"
        "v_mad_u64_u32 %0, s[8:9], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[4:5]\n\t"
        "v_mad_u64_u32 %0, s[4:5], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[6:7]\n\t"
        "v_mad_u64_u32 %0, s[6:7], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[8:9]\n\t"
        "v_mad_u64_u32 %0, s[8:9], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[4:5]\n\t"
        "v_mad_u64_u32 %0, s[4:5], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[6:7]\n\t"
        "v_mad_u64_u32 %0, s[6:7], %2, %3, %0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[8:9]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[4:5]\n\t"
        "v_addc_co_u32 %1, s[10:11], 0, %1, s[6:7]\n\t"
        : "+v"(acc), "+v"(h) : "v"(a), "v"(b) : "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ h;
}

// v_mul_u32_u24 + v_mul_hi_u32_u24
__global__ void k_mul24pair(u64 *out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mul_hi_u32_u24 %0, %0, %8\n\t"
        "v_mul_hi_u32_u24 %1, %1, %8\n\t"
        "v_mul_hi_u32_u24 %2, %2, %8\n\t"
        "v_mul_hi_u32_u24 %3, %3, %8\n\t"
        "v_mul_hi_u32_u24 %4, %4, %8\n\t"
        "v_mul_hi_u32_u24 %5, %5, %8\n\t"
        "v_mul_hi_u32_u24 %6, %6, %8\n\t"
        "v_mul_hi_u32_u24 %7, %7, %8\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

// f64 fma independent
__global__ void k_fma64(u64 *out, u32 seed) {
  double a = 1.0 + 1e-9 * (threadIdx.x + seed);
  double c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_fma_f64 %0, %0, %8, %0\n\t"
        "v_fma_f64 %1, %1, %8, %1\n\t"
        "v_fma_f64 %2, %2, %8, %2\n\t"
        "v_fma_f64 %3, %3, %8, %3\n\t"
        "v_fma_f64 %4, %4, %8, %4\n\t"
        "v_fma_f64 %5, %5, %8, %5\n\t"
        "v_fma_f64 %6, %6, %8, %6\n\t"
        "v_fma_f64 %7, %7, %8, %7\n\t"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
}

typedef void (*kern_t)(u64 *, u32);

static void run(const char *name, kern_t k, int instr_per_iter, int waves_per_simd, u64 *dout, int ncu) {
  // one block per CU with 256*waves_per_simd threads => waves_per_simd waves on each SIMD
  int threads = 256 * waves_per_simd;
  int blocks = ncu;
  if (threads > 1024) { blocks = ncu * (threads / 1024); threads = 1024; }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, 1u);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, (u32)rep);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double total_wave_instr = (double)blocks * (threads / 64) * (double)ITERS * instr_per_iter;
  double lane_ops = total_wave_instr * 64.0;
  // cycles per wave-instruction per SIMD assuming 2.4 GHz
  double simds = ncu * 4.0;
  double cyc = best * 1e-3 * 2.4e9 * simds / total_wave_instr;
  printf("%-18s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-ops/s  %6.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name, waves_per_simd, best,
         lane_ops / (best * 1e-3) / 1e12, cyc);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount;
  u64 *dout; CHECK(hipMalloc(&dout, (size_t)ncu * 8 * 1024 * 8));
  int ws[] = {1, 2, 4, 8};
  for (int wi = 0; wi < 4; ++wi) {
    int w = ws[wi];
    run("mad_u64 indep8", k_mad_indep8, 8, w, dout, ncu);
    run("mad_u64 dep", k_mad_dep, 8, w, dout, ncu);
    run("mul_lo_u32", k_mullo, 8, w, dout, ncu);
    run("mul_hi_u32", k_mulhi, 8, w, dout, ncu);
    run("mad_u32_u24", k_mad24, 8, w, dout, ncu);
    run("mul_hi_u32_u24", k_mul24pair, 8, w, dout, ncu);
    run("add_u32", k_add, 8, w, dout, ncu);
    run("lshl_add_u64", k_add64, 8, w, dout, ncu);
    run("fma_f64", k_fma64, 8, w, dout, ncu);
    run("carry8 +s_nop1", k_carry_nop, 8, w, dout, ncu);
    run("carry8 no nop", k_carry_nonop, 8, w, dout, ncu);
    run("mad+addc rot(8+8)", k_mad_addc_rot, 16, w, dout, ncu);
    printf("\n");
  }
  return 0;
}
