// generated: single-instruction VALU throughput table for gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define ITERS 16384


__global__ void k0(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k1(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_add_co_u32 %0, vcc, %0, %8\n\tv_add_co_u32 %1, vcc, %1, %8\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_add_co_u32 %3, vcc, %3, %8\n\tv_add_co_u32 %4, vcc, %4, %8\n\tv_add_co_u32 %5, vcc, %5, %8\n\tv_add_co_u32 %6, vcc, %6, %8\n\tv_add_co_u32 %7, vcc, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k2(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_add_co_u32 %0, s[40:41], %0, %8\n\tv_add_co_u32 %1, s[42:43], %1, %8\n\tv_add_co_u32 %2, s[44:45], %2, %8\n\tv_add_co_u32 %3, s[46:47], %3, %8\n\tv_add_co_u32 %4, s[48:49], %4, %8\n\tv_add_co_u32 %5, s[50:51], %5, %8\n\tv_add_co_u32 %6, s[52:53], %6, %8\n\tv_add_co_u32 %7, s[54:55], %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k3(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\tv_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k4(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_addc_co_u32 %0, s[40:41], %0, %8, s[40:41]\n\tv_addc_co_u32 %1, s[42:43], %1, %8, s[42:43]\n\tv_addc_co_u32 %2, s[44:45], %2, %8, s[44:45]\n\tv_addc_co_u32 %3, s[46:47], %3, %8, s[46:47]\n\tv_addc_co_u32 %4, s[48:49], %4, %8, s[48:49]\n\tv_addc_co_u32 %5, s[50:51], %5, %8, s[50:51]\n\tv_addc_co_u32 %6, s[52:53], %6, %8, s[52:53]\n\tv_addc_co_u32 %7, s[54:55], %7, %8, s[54:55]"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k5(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k6(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k7(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k8(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %1, 3, %1\n\tv_lshrrev_b32 %2, 3, %2\n\tv_lshrrev_b32 %3, 3, %3\n\tv_lshrrev_b32 %4, 3, %4\n\tv_lshrrev_b32 %5, 3, %5\n\tv_lshrrev_b32 %6, 3, %6\n\tv_lshrrev_b32 %7, 3, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k9(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_alignbit_b32 %0, %0, %8, 30\n\tv_alignbit_b32 %1, %1, %8, 30\n\tv_alignbit_b32 %2, %2, %8, 30\n\tv_alignbit_b32 %3, %3, %8, 30\n\tv_alignbit_b32 %4, %4, %8, 30\n\tv_alignbit_b32 %5, %5, %8, 30\n\tv_alignbit_b32 %6, %6, %8, 30\n\tv_alignbit_b32 %7, %7, %8, 30"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k10(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k11(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_add3_u32 %0, %0, %8, %8\n\tv_add3_u32 %1, %1, %8, %8\n\tv_add3_u32 %2, %2, %8, %8\n\tv_add3_u32 %3, %3, %8, %8\n\tv_add3_u32 %4, %4, %8, %8\n\tv_add3_u32 %5, %5, %8, %8\n\tv_add3_u32 %6, %6, %8, %8\n\tv_add3_u32 %7, %7, %8, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k12(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n\tv_lshl_add_u32 %1, %1, 2, %8\n\tv_lshl_add_u32 %2, %2, 2, %8\n\tv_lshl_add_u32 %3, %3, 2, %8\n\tv_lshl_add_u32 %4, %4, 2, %8\n\tv_lshl_add_u32 %5, %5, 2, %8\n\tv_lshl_add_u32 %6, %6, 2, %8\n\tv_lshl_add_u32 %7, %7, 2, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k13(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k14(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k15(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_cmp_lt_u32 s[40:41], %0, %8\n\tv_cmp_lt_u32 s[42:43], %1, %8\n\tv_cmp_lt_u32 s[44:45], %2, %8\n\tv_cmp_lt_u32 s[46:47], %3, %8\n\tv_cmp_lt_u32 s[48:49], %4, %8\n\tv_cmp_lt_u32 s[50:51], %5, %8\n\tv_cmp_lt_u32 s[52:53], %6, %8\n\tv_cmp_lt_u32 s[54:55], %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k16(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k17(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mad_u64_u32 %0, s[40:41], %8, %8, %0\n\tv_mad_u64_u32 %1, s[42:43], %8, %8, %1\n\tv_mad_u64_u32 %2, s[44:45], %8, %8, %2\n\tv_mad_u64_u32 %3, s[46:47], %8, %8, %3\n\tv_mad_u64_u32 %4, s[48:49], %8, %8, %4\n\tv_mad_u64_u32 %5, s[50:51], %8, %8, %5\n\tv_mad_u64_u32 %6, s[52:53], %8, %8, %6\n\tv_mad_u64_u32 %7, s[54:55], %8, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k18(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n\tv_mad_u64_u32 %1, vcc, %8, %8, %1\n\tv_mad_u64_u32 %2, vcc, %8, %8, %2\n\tv_mad_u64_u32 %3, vcc, %8, %8, %3\n\tv_mad_u64_u32 %4, vcc, %8, %8, %4\n\tv_mad_u64_u32 %5, vcc, %8, %8, %5\n\tv_mad_u64_u32 %6, vcc, %8, %8, %6\n\tv_mad_u64_u32 %7, vcc, %8, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k19(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mad_i64_i32 %0, s[40:41], %8, %8, %0\n\tv_mad_i64_i32 %1, s[42:43], %8, %8, %1\n\tv_mad_i64_i32 %2, s[44:45], %8, %8, %2\n\tv_mad_i64_i32 %3, s[46:47], %8, %8, %3\n\tv_mad_i64_i32 %4, s[48:49], %8, %8, %4\n\tv_mad_i64_i32 %5, s[50:51], %8, %8, %5\n\tv_mad_i64_i32 %6, s[52:53], %8, %8, %6\n\tv_mad_i64_i32 %7, s[54:55], %8, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k20(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_lshl_add_u64 %0, %0, 0, %0\n\tv_lshl_add_u64 %1, %1, 0, %1\n\tv_lshl_add_u64 %2, %2, 0, %2\n\tv_lshl_add_u64 %3, %3, 0, %3\n\tv_lshl_add_u64 %4, %4, 0, %4\n\tv_lshl_add_u64 %5, %5, 0, %5\n\tv_lshl_add_u64 %6, %6, 0, %6\n\tv_lshl_add_u64 %7, %7, 0, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k21(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_lshrrev_b64 %0, 30, %0\n\tv_lshrrev_b64 %1, 30, %1\n\tv_lshrrev_b64 %2, 30, %2\n\tv_lshrrev_b64 %3, 30, %3\n\tv_lshrrev_b64 %4, 30, %4\n\tv_lshrrev_b64 %5, 30, %5\n\tv_lshrrev_b64 %6, 30, %6\n\tv_lshrrev_b64 %7, 30, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k22(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_lshlrev_b64 %0, 2, %0\n\tv_lshlrev_b64 %1, 2, %1\n\tv_lshlrev_b64 %2, 2, %2\n\tv_lshlrev_b64 %3, 2, %3\n\tv_lshlrev_b64 %4, 2, %4\n\tv_lshlrev_b64 %5, 2, %5\n\tv_lshlrev_b64 %6, 2, %6\n\tv_lshlrev_b64 %7, 2, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k23(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_ashrrev_i64 %0, 30, %0\n\tv_ashrrev_i64 %1, 30, %1\n\tv_ashrrev_i64 %2, 30, %2\n\tv_ashrrev_i64 %3, 30, %3\n\tv_ashrrev_i64 %4, 30, %4\n\tv_ashrrev_i64 %5, 30, %5\n\tv_ashrrev_i64 %6, 30, %6\n\tv_ashrrev_i64 %7, 30, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k24(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k25(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mul_hi_u32 %0, %0, %8\n\tv_mul_hi_u32 %1, %1, %8\n\tv_mul_hi_u32 %2, %2, %8\n\tv_mul_hi_u32 %3, %3, %8\n\tv_mul_hi_u32 %4, %4, %8\n\tv_mul_hi_u32 %5, %5, %8\n\tv_mul_hi_u32 %6, %6, %8\n\tv_mul_hi_u32 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k26(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n\tv_mad_u32_u24 %1, %1, %8, %1\n\tv_mad_u32_u24 %2, %2, %8, %2\n\tv_mad_u32_u24 %3, %3, %8, %3\n\tv_mad_u32_u24 %4, %4, %8, %4\n\tv_mad_u32_u24 %5, %5, %8, %5\n\tv_mad_u32_u24 %6, %6, %8, %6\n\tv_mad_u32_u24 %7, %7, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k27(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mul_u32_u24 %0, %0, %8\n\tv_mul_u32_u24 %1, %1, %8\n\tv_mul_u32_u24 %2, %2, %8\n\tv_mul_u32_u24 %3, %3, %8\n\tv_mul_u32_u24 %4, %4, %8\n\tv_mul_u32_u24 %5, %5, %8\n\tv_mul_u32_u24 %6, %6, %8\n\tv_mul_u32_u24 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k28(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mad_u32_u16 %0, %0, %8, %0\n\tv_mad_u32_u16 %1, %1, %8, %1\n\tv_mad_u32_u16 %2, %2, %8, %2\n\tv_mad_u32_u16 %3, %3, %8, %3\n\tv_mad_u32_u16 %4, %4, %8, %4\n\tv_mad_u32_u16 %5, %5, %8, %5\n\tv_mad_u32_u16 %6, %6, %8, %6\n\tv_mad_u32_u16 %7, %7, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k29(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\tv_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k30(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\tv_fma_f64 %4, %4, %4, %4\n\tv_fma_f64 %5, %5, %5, %5\n\tv_fma_f64 %6, %6, %6, %6\n\tv_fma_f64 %7, %7, %7, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k31(u64 *out, u32 seed, u64 *clk) {
  u64 a = (u64)(threadIdx.x * 2654435761u + seed);
  u64 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3\n\tv_pk_fma_f32 %4, %4, %4, %4\n\tv_pk_fma_f32 %5, %5, %5, %5\n\tv_pk_fma_f32 %6, %6, %6, %6\n\tv_pk_fma_f32 %7, %7, %7, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k32(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k33(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_dot4_i32_i8 %0, %0, %8, %0\n\tv_dot4_i32_i8 %1, %1, %8, %1\n\tv_dot4_i32_i8 %2, %2, %8, %2\n\tv_dot4_i32_i8 %3, %3, %8, %3\n\tv_dot4_i32_i8 %4, %4, %8, %4\n\tv_dot4_i32_i8 %5, %5, %8, %5\n\tv_dot4_i32_i8 %6, %6, %8, %6\n\tv_dot4_i32_i8 %7, %7, %8, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k34(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_write_b32 a3, %3\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_write_b32 a4, %4\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_write_b32 a5, %5\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_write_b32 a6, %6\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_write_b32 a7, %7\n\tv_accvgpr_read_b32 %7, a7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc","a0","a1","a2","a3","a4","a5","a6","a7");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k35(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\tds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\ts_waitcnt lgkmcnt(0)"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k36(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void k37(u64 *out, u32 seed, u64 *clk) {
  u32 a = (u32)(threadIdx.x * 2654435761u + seed);
  u32 c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
  u32 b = (u32)a | 1;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_readlane_b32 s40, %0, 3\n\tv_readlane_b32 s42, %1, 3\n\tv_readlane_b32 s44, %2, 3\n\tv_readlane_b32 s46, %3, 3\n\tv_readlane_b32 s48, %4, 3\n\tv_readlane_b32 s50, %5, 3\n\tv_readlane_b32 s52, %6, 3\n\tv_readlane_b32 s54, %7, 3"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
      : "v"(b) : "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc");
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef void (*kern_t)(u64 *, u32, u64 *);
static void run(const char *name, kern_t k, int instr_per_iter, int waves_per_simd, u64 *dout, u64 *dclk, int ncu) {
  int threads = 256 * waves_per_simd; int blocks = ncu;
  if (threads > 1024) { blocks = ncu * (threads / 1024); threads = 1024; }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, 1u, dclk);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, (u32)rep, dclk);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  u64 clk; CHECK(hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost));
  double wave_instr_per_simd = (double)waves_per_simd * ITERS * instr_per_iter;
  // s_memtime counts at a fixed 100MHz? report both
  printf("%-20s w/SIMD=%d %8.3f ms  wall-cyc/instr@2.4GHz=%6.2f  memtime-ticks/instr(wave0)=%7.3f\n", name, waves_per_simd, best,
         best * 1e-3 * 2.4e9 / wave_instr_per_simd, (double)clk / ((double)ITERS * instr_per_iter));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}
int main(int argc, char **argv) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  u64 *dout, *dclk; CHECK(hipMalloc(&dout, (size_t)ncu * 8 * 1024 * 8)); CHECK(hipMalloc(&dclk, 64));
  int ws[] = {1, 2, 4};
  for (int wi = 0; wi < 3; ++wi) {

    run("v_add_u32", k0, 8, ws[wi], dout, dclk, ncu);
    run("v_add_co_u32 vcc", k1, 8, ws[wi], dout, dclk, ncu);
    run("v_add_co_u32 sgpr", k2, 8, ws[wi], dout, dclk, ncu);
    run("v_addc_co_u32 vcc", k3, 8, ws[wi], dout, dclk, ncu);
    run("v_addc sgprs", k4, 8, ws[wi], dout, dclk, ncu);
    run("v_sub_u32", k5, 8, ws[wi], dout, dclk, ncu);
    run("v_and_b32", k6, 8, ws[wi], dout, dclk, ncu);
    run("v_xor_b32", k7, 8, ws[wi], dout, dclk, ncu);
    run("v_lshrrev_b32", k8, 8, ws[wi], dout, dclk, ncu);
    run("v_alignbit_b32", k9, 8, ws[wi], dout, dclk, ncu);
    run("v_and_or_b32", k10, 8, ws[wi], dout, dclk, ncu);
    run("v_add3_u32", k11, 8, ws[wi], dout, dclk, ncu);
    run("v_lshl_add_u32", k12, 8, ws[wi], dout, dclk, ncu);
    run("v_cndmask vcc", k13, 8, ws[wi], dout, dclk, ncu);
    run("v_cmp_lt_u32 vcc", k14, 8, ws[wi], dout, dclk, ncu);
    run("v_cmp_lt_u32 sgpr", k15, 8, ws[wi], dout, dclk, ncu);
    run("v_mov_b32", k16, 8, ws[wi], dout, dclk, ncu);
    run("v_mad_u64_u32", k17, 8, ws[wi], dout, dclk, ncu);
    run("v_mad_u64_u32 vcc", k18, 8, ws[wi], dout, dclk, ncu);
    run("v_mad_i64_i32", k19, 8, ws[wi], dout, dclk, ncu);
    run("v_lshl_add_u64", k20, 8, ws[wi], dout, dclk, ncu);
    run("v_lshrrev_b64", k21, 8, ws[wi], dout, dclk, ncu);
    run("v_lshlrev_b64", k22, 8, ws[wi], dout, dclk, ncu);
    run("v_ashrrev_i64", k23, 8, ws[wi], dout, dclk, ncu);
    run("v_mul_lo_u32", k24, 8, ws[wi], dout, dclk, ncu);
    run("v_mul_hi_u32", k25, 8, ws[wi], dout, dclk, ncu);
    run("v_mad_u32_u24", k26, 8, ws[wi], dout, dclk, ncu);
    run("v_mul_u32_u24", k27, 8, ws[wi], dout, dclk, ncu);
    run("v_mad_u32_u16", k28, 8, ws[wi], dout, dclk, ncu);
    run("v_fma_f32", k29, 8, ws[wi], dout, dclk, ncu);
    run("v_fma_f64", k30, 8, ws[wi], dout, dclk, ncu);
    run("v_pk_fma_f32", k31, 8, ws[wi], dout, dclk, ncu);
    run("v_pk_add_u16", k32, 8, ws[wi], dout, dclk, ncu);
    run("v_dot4_i32_i8", k33, 8, ws[wi], dout, dclk, ncu);
    run("v_accvgpr_rw", k34, 16, ws[wi], dout, dclk, ncu);
    run("ds_bpermute", k35, 8, ws[wi], dout, dclk, ncu);
    run("v_mov_dpp", k36, 8, ws[wi], dout, dclk, ncu);
    run("v_readlane s", k37, 8, ws[wi], dout, dclk, ncu);
    printf("\n");
  }
  return 0;
}
